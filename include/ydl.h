/*
 * ydl.h — C ABI of libydl_hip.so: the MI355X (gfx950) kernels behind the segmentation training hot path.
 *
 * Drop-in boundary.  The reference has no native ABI on its live path (it calls ATen through
 * nn.Module.forward); the only native interface it ships is the orphan DCNv3 pybind pair
 * (models/ops_dcnv3/src/cuda/dcnv3_cuda.h:15-31).  This header is what a reference-side binding (ctypes /
 * pybind) would bind to replace, per op, the ATen calls made by:
 *   Conv.forward            unet-lite/yolo5-seg/seg_diceloss_yolov5.py:403-409  (models/common.py:57-64)
 *   C3/C2f/SPPF/Concat      seg_diceloss_yolov5.py:416-507, yolov8/seg_jaccardloss_yolov8.py:401-414
 *   nn.Upsample / interpolate  seg_diceloss_yolov5.py:588-609, 655-657
 *   SegmentHead             segment/train.py:159-210
 *   SegmentationLoss        seg_diceloss_yolov5.py:712-750, yolov8/seg_jaccardloss_yolov8.py:774-815
 *   smart_optimizer + ModelEMA.update   utils/torch_utils.py:318-346, 404-428
 *   dcnv3_forward/backward  models/ops_dcnv3/src/cuda/dcnv3_cuda.h:15-31
 *
 * Conventions
 *  - Plain pointers and sizes only; all pointers are DEVICE pointers unless named h_*.
 *  - The caller owns every buffer (outputs, workspaces); kernels never allocate or synchronise.
 *  - Every entry point takes the HIP stream explicitly (void* = hipStream_t) and is re-entrant.
 *  - Return value: 0 on success, non-zero on error; ydl_last_error() gives a thread-local message.
 *  - Activations are NHWC: element (n,h,w,c) of a tensor with pixel stride `ld` lives at
 *    base[((n*H + h)*W + w)*ld + c].  `ld >= C`, `ld*sizeof(elem)` and `base` must be 16-byte aligned.
 *    A channel slice of a wider buffer (free concat) is just base+c0 with the wide ld.
 *  - dtype: YDL_F32 (exact-parity mode, f32 MFMA) or YDL_BF16 (throughput mode, bf16 MFMA, f32 accumulate).
 */
#ifndef YDL_H_
#define YDL_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { YDL_F32 = 0, YDL_BF16 = 1, YDL_F16 = 2 /* DCNv3 op only */ };
enum { YDL_ACT_NONE = 0, YDL_ACT_SILU = 1, YDL_ACT_RELU = 2 };
/* residual handling of the fused BN/activation kernels */
enum { YDL_RES_NONE = 0, YDL_RES_AFTER_ACT = 1 /* C3/C2f: act(bn(y)) + r */, YDL_RES_BEFORE_ACT = 2 /* ResNet: act(bn(y) + r) */,
       YDL_RES_GRAD_ACCUMULATE = 16 /* ydl_bn_act_bwd only, or-ed into res_mode: dres += instead of dres = */ };
enum { YDL_LOSS_DICE = 0, YDL_LOSS_JACCARD = 1 };

const char* ydl_last_error(void);
int ydl_version(void);
/* Test-only, process-wide debug knobs for A/B tests (key 0: bf16 wgrad operand path, 1: streaming point-wise kernel on/off,
 * 2: strided dgrad in one launch / per class).  They change launch geometry: a caller that caches ydl_conv_fwd_grid_m /
 * _block_m / _stats_ws_bytes must drop its cache after a change.  Everything else in the library is per call or per
 * device (kernel attributes and the CU count are keyed by the HIP device id current at the call). */
void ydl_debug_set(int key, int val);
/* diagnostics: number of (kernel, device) launch-attribute initialisations done so far */
int ydl_debug_attr_sets(void);
/* diagnostics: name of the kernel instantiation the last call of an entry family launched (process-wide);
 * family 0 ydl_conv_fwd, 1 ydl_conv_dgrad, 2 ydl_conv_wgrad, 3 ydl_bn_finalize.  "" if none yet. */
const char* ydl_debug_last_kernel(int family);

/* ---- convolution as implicit GEMM on MFMA ------------------------------------------------------------
 * Geometry of one conv layer (square kernel k, stride s, padding p, groups=1, no bias).
 * Weights in compute layout: w  = [Cout][k*k][Cin_p]   (KRSC; Cin_p = Cin rounded up to 8)
 *                            wt = [Cin][k*k][Cout_p]   (transposed, for dgrad; Cout_p = Cout rounded up to 8)
 */
typedef struct {
    int N, Hi, Wi, Cin;       /* input  (Cin = logical channels; reads Cin_p = round_up(Cin, 8) <= ldx) */
    int Ho, Wo, Cout;         /* output */
    int k, s, p;
    int ldx, ldy;             /* pixel strides of x and y (elements) */
    int ldw;                  /* row stride (elements) of w in conv_fwd and of dw in conv_wgrad; 0 = dense = k*k*Cin_p.
                                 A larger stride addresses a COLUMN BLOCK of a wider 1x1 weight matrix: a convolution
                                 over a channel concat is evaluated as the sum of one convolution per source */
} ydl_conv_geom;

/* BN-statistics workspace of the forward epilogue: [grid_m][2][round_up(Cout,8)] floats, where block b of
 * the launch covered min(block_m, N*Ho*Wo - b*block_m) pixels.  The three queries are pure functions of g. */
int64_t ydl_conv_fwd_stats_ws_bytes(const ydl_conv_geom* g, int dtype);
int ydl_conv_fwd_grid_m(const ydl_conv_geom* g, int dtype);
int ydl_conv_fwd_block_m(const ydl_conv_geom* g, int dtype);

/* y (+)= conv(x, w).  If stats_ws != NULL the epilogue also writes per-block (sum, M2) partials of the values it
 * stores (f32 accumulators, plus the previous contents of y when accumulate != 0) per output channel for train-mode
 * BN; finish them with ydl_bn_finalize. */
int ydl_conv_fwd(const ydl_conv_geom* g, int dtype, const void* x, const void* w, void* y,
                 float* stats_ws, int accumulate, void* stream);
/* dx (+)= conv_transpose(dy, wt).  accumulate != 0 adds into dx (gradient fan-in). */
int ydl_conv_dgrad(const ydl_conv_geom* g, int dtype, const void* dy, const void* wt, void* dx,
                   int accumulate, void* stream);
/* Input gradient AND weight gradient of a 1x1 / stride-1 convolution in ONE pass over dy (both are HBM-bound on large maps: dy is
 * fetched once instead of twice; csrc/igemm.hip: pwbw_kernel).  Supported (ydl_conv_bwd_pw_supported != 0): bf16, Cin = Cout = 128,
 * at least 131 072 pixels.  dx has its own pixel stride lddx (a channel slice of a wider gradient buffer); accumulate != 0 adds into
 * dx; dw[Cout][g->ldw or 128] receives f32 atomic adds like ydl_conv_wgrad.  wt = [128][128] (ydl_conv_dgrad's operand). */
int ydl_conv_bwd_pw_supported(const ydl_conv_geom* g, int dtype);
int ydl_conv_bwd_pw(const ydl_conv_geom* g, int dtype, const void* x, const void* dy, const void* wt, void* dx, int lddx,
                    int accumulate, float* dw, void* stream);
/* The input gradient with the BatchNorm backward's REDUCE pass of the layer(s) that produced the convolution's input fused into
 * its epilogue (throughput mode: replica sums, see ydl_bn_act_bwd_sums).  dx — the gradient this call completes, i.e. the `dout` of
 * those layers — is still in registers when it is stored: the epilogue reads the producers' saved pre-activations y once,
 * forms dz = dx * act'(y*scale + shift) and adds (sum dz, sum dz*xhat) per channel into sums[i] = [YDL_BN_REPLICAS][2][cp[i]]
 * (zeroed by the caller), so ydl_bn_act_bwd_apply_sums can follow without the reduce launch and its second read of dx.
 * Valid only when this call is the LAST writer of dx (accumulate != 0 is fine: the stored sum is what is reduced).
 * Up to two channel segments [c0, c1) of dx (multiples of 8), each with its own producer: a channel concat of two layers.
 * Per-segment pointers are at the segment's channel 0 (= channel c0 of dx).  act: YDL_ACT_NONE or YDL_ACT_SILU.
 * ydl_conv_dgrad_bnred_supported: 1 when this geometry runs on a kernel whose epilogue has the fused form (bf16 ring kernels),
 * else use ydl_conv_dgrad + ydl_bn_act_bwd_sums. */
typedef struct {
    int nseg;
    int c0[2], c1[2], ldy[2], cp[2], act[2];
    const void* y[2];
    const float* scale[2];
    const float* shift[2];
    const float* mean[2];
    const float* invstd[2];
    float* sums[2];
} ydl_bnred;
int ydl_conv_dgrad_bnred_supported(const ydl_conv_geom* g, int dtype);
int ydl_conv_dgrad_bnred(const ydl_conv_geom* g, int dtype, const void* dy, const void* wt, void* dx,
                         int accumulate, const ydl_bnred* red, void* stream);
/* dw[Cout][k*k][Cin_p] (f32) += sum over pixels dy^T * im2col(x).  dw must be zeroed (or hold the running
 * sum for gradient accumulation) before the call: split-K blocks add with f32 atomics. */
int ydl_conv_wgrad(const ydl_conv_geom* g, int dtype, const void* x, const void* dy, float* dw, void* stream);
/* Deterministic form (bitwise reproducible from run to run): every split-K block stores its partial tile into
 * ws = [splits][Cout][k*k*Cin_p] f32 (ydl_conv_wgrad_ws_bytes) and a second kernel adds the partials into dw in a fixed
 * order.  Same products and tiles as ydl_conv_wgrad; parity mode (YDL_F32) uses it by default. */
int64_t ydl_conv_wgrad_ws_bytes(const ydl_conv_geom* g, int dtype);
int ydl_conv_wgrad_det(const ydl_conv_geom* g, int dtype, const void* x, const void* dy, float* dw, float* ws, void* stream);

/* Throughput-mode statistics: the same convolution, but every block ADDS its per-channel (sum, sum of squares) with f32 atomics
 * into sums = [YDL_BN_REPLICAS][2][round_up(Cout,8)] floats (replica = block index mod YDL_BN_REPLICAS, which spreads the
 * contention; the caller zeroes the buffer) instead of writing a partial row: no ydl_bn_finalize launch — ydl_bn_act_fwd_sums
 * derives the coefficients from the sums itself.  The sums depend on arrival order in their last bits (run-to-run differences
 * of 1e-7 relative in mean / variance): parity mode keeps the deterministic partial rows + ydl_bn_finalize. */
#define YDL_BN_REPLICAS 8
int ydl_conv_fwd_sums(const ydl_conv_geom* g, int dtype, const void* x, const void* w, void* y,
                      float* sums, int accumulate, void* stream);

/* master weights (f32, KRSC [Cout][k*k][Cin]) -> compute copies: w [Cout][kk][Cin_p] and wt [Cin][kk][Cout_p] */
int ydl_weight_prep(int dtype, const float* master, void* w, void* wt, int Cout, int kk, int Cin, void* stream);
/* the same for every layer of a model in one launch; desc_dev: device array of nlayers x 8 int64
 * {master*, w*, wt*, Cout, kk, Cin, 0, 0} */
int ydl_weight_prep_batched(int dtype, const int64_t* desc_dev, int nlayers, void* stream);
/* dw [Cout][kk][Cin_p] f32 -> master-layout grad [Cout][kk][Cin] (only needed when Cin_p != Cin) */
int ydl_wgrad_unpad(const float* dw, float* grad, int Cout, int kk, int Cin, int accumulate, void* stream);

/* ---- train-mode BatchNorm + activation ------------------------------------------------------------- */
/* Chan-merge the per-block partials; writes mean/invstd (saved for backward), scale=gamma*invstd,
 * shift=beta-mean*scale; updates running_mean/var (momentum, unbiased var) and is a no-op on them if NULL. */
int ydl_bn_finalize(const float* stats_ws, int nblocks, int block_m, int64_t count, int C,
                    const float* gamma, const float* beta,
                    float eps, float momentum, float* running_mean, float* running_var,
                    float* mean, float* invstd, float* scale, float* shift,
                    int replication /* >=1: the logical tensor is the stored one nearest-replicated this many times
                                       (lazy up-sampling); only the unbiased running_var factor depends on it */,
                    void* stream);
/* eval-mode: scale/shift from running statistics */
int ydl_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift, void* stream);
/* out = act(y*scale+shift) [+ res] ; res_mode per YDL_RES_*.  Cp = channels processed (multiple of 8). */
int ydl_bn_act_fwd(int dtype, const void* y, int ldy, const float* scale, const float* shift,
                   const void* res, int ldr, int res_mode, int act, void* out, int ldo,
                   int64_t npix, int Cp, void* stream);
int64_t ydl_bn_bwd_ws_bytes(int64_t npix, int Cp);
/* Backward of out = act(bn(y)) [+res].  dout: grad wrt out.  out: saved output (needed only for RELU).
 * Writes dy (grad wrt conv output), dgamma/dbeta (f32, accumulate flag) and, when dres is not NULL, the gradient of the residual
 * branch in the same pass: the masked gradient dz for YDL_RES_BEFORE_ACT, dout itself for YDL_RES_AFTER_ACT; res_mode may carry
 * YDL_RES_GRAD_ACCUMULATE (dres already holds another consumer's gradient: add to it). */
int ydl_bn_act_bwd(int dtype, const void* y, int ldy, const void* dout, int lddo, const void* out, int ldo,
                   const float* gamma, const float* mean, const float* invstd, const float* scale, const float* shift,
                   int res_mode, int act, void* dy, int lddy, void* dres, int lddr,
                   float* dgamma, float* dbeta, int accumulate_param_grads,
                   float* ws, int64_t npix, int C, int Cp, void* stream);

/* The two-launch forms on replica sums (throughput mode; see ydl_conv_fwd_sums).
 * ydl_bn_act_fwd_sums = ydl_bn_finalize + ydl_bn_act_fwd in ONE launch: every thread derives mean / invstd / scale / shift of its
 * channel chunk from sums = [YDL_BN_REPLICAS][2][sums_ld] (count = pixels the sums cover), block 0 also stores them (mean, invstd,
 * scale, shift: [Cp] each, read again by the backward) and updates the running statistics.  All per-channel pointers and `sums`
 * may point INTO wider arrays (a channel group of a fused sibling convolution): sums_ld is the row stride of the sums.
 * ydl_bn_act_bwd_sums = ydl_bn_act_bwd without its merge launch: the reduce pass adds (sum dz, sum dz*xhat) into
 * sums = [YDL_BN_REPLICAS][2][Cp] (zeroed by the caller), the apply pass reads them; block 0 of the apply pass stores or adds
 * dgamma / dbeta. */
int ydl_bn_act_fwd_sums(int dtype, const void* y, int ldy, const float* sums, int sums_ld, int64_t count,
                        const float* gamma, const float* beta, float eps, float momentum,
                        float* running_mean, float* running_var, float* mean, float* invstd, float* scale, float* shift,
                        int replication, const void* res, int ldr, int res_mode, int act, void* out, int ldo,
                        int64_t npix, int C, int Cp, void* stream);
int ydl_bn_act_bwd_sums(int dtype, const void* y, int ldy, const void* dout, int lddo, const void* out, int ldo,
                        const float* mean, const float* invstd, const float* scale, const float* shift,
                        int res_mode, int act, void* dy, int lddy, void* dres, int lddr,
                        float* dgamma, float* dbeta, int accumulate_param_grads,
                        float* sums, int64_t npix, int C, int Cp, void* stream);
/* the apply pass alone: sums already hold (sum dz, sum dz*xhat) of this tensor (ydl_conv_dgrad_bnred) */
int ydl_bn_act_bwd_apply_sums(int dtype, const void* y, int ldy, const void* dout, int lddo, const void* out, int ldo,
                              const float* mean, const float* invstd, const float* scale, const float* shift,
                              int res_mode, int act, void* dy, int lddy, void* dres, int lddr,
                              float* dgamma, float* dbeta, int accumulate_param_grads,
                              float* sums, int64_t npix, int C, int Cp, void* stream);

/* ---- spatial ops (NHWC, channel-vectorised) -------------------------------------------------------- */
/* max pool (k,s,p), -inf padding; idx (uint8 window offset of the arg-max, first max in scan order) is
 * written when non-NULL and consumed by the backward. */
int ydl_maxpool_fwd(int dtype, const void* x, int ldx, void* y, int ldy, uint8_t* idx,
                    int N, int Hi, int Wi, int Ho, int Wo, int C, int k, int s, int p, void* stream);
int ydl_maxpool_bwd(int dtype, const void* dy, int lddy, const uint8_t* idx, void* dx, int lddx, int accumulate,
                    int N, int Hi, int Wi, int Ho, int Wo, int C, int k, int s, int p, void* stream);
/* SPPF's chain of three k x k / stride 1 / pad k/2 max-pools (y1 = mp(x), y2 = mp(y1), y3 = mp(y2); seg_diceloss_yolov5.py:468-481,
 * models/common.py:223-238) in one launch per direction, the H x W plane held in LDS: same values, index planes and gradients as
 * three ydl_maxpool_fwd / ydl_maxpool_bwd calls, bit for bit.  ydl_sppf_pool_supported: 1 when the plane fits (else use those).
 * Backward: dyK = gradient already present in the slice of yK (the concat consumer's input gradient), dx (+)= the chain's result. */
int ydl_sppf_pool_supported(int dtype, int H, int W, int C, int k);
int ydl_sppf_pool_fwd(int dtype, const void* x, int ldx, void* y1, void* y2, void* y3, int ldy,
                      uint8_t* idx1, uint8_t* idx2, uint8_t* idx3, int N, int H, int W, int C, int k, void* stream);
int ydl_sppf_pool_bwd(int dtype, const void* dy1, const void* dy2, const void* dy3, int lddy, const uint8_t* idx1,
                      const uint8_t* idx2, const uint8_t* idx3, void* dx, int lddx, int accumulate,
                      int N, int H, int W, int C, int k, void* stream);
/* resize: mode 0 nearest (src=min(floor(dst*scale),in-1)), 1 bilinear align_corners=False, 2 bilinear
 * align_corners=True.  scale_h/w <= 0 means "derive from sizes" (in/out, or (in-1)/(out-1)). */
int ydl_resize_fwd(int dtype, int mode, const void* x, int ldx, void* y, int ldy,
                   int N, int Hi, int Wi, int Ho, int Wo, int C, float scale_h, float scale_w, void* stream);
int ydl_resize_bwd(int dtype, int mode, const void* dy, int lddy, void* dx, int lddx, int accumulate,
                   int N, int Hi, int Wi, int Ho, int Wo, int C, float scale_h, float scale_w, void* stream);
/* y += resize(x) (same modes and index arithmetic as ydl_resize_fwd); with ``sums`` != NULL also adds the per-channel (sum, sum of
 * squares) of the result to BatchNorm replica rows sums[YDL_BN_REPLICAS][2][sums_ld], exactly like ydl_conv_fwd_sums' epilogues: the
 * last launch of a 1x1 Conv over the auto-aligned Concat [a, resize(b)] (seg_diceloss_yolov5.py:484-507 + :388-409) evaluated as
 * conv_a(a) + resize(conv_b(b)) */
int ydl_resize_acc_sums(int dtype, int mode, const void* x, int ldx, void* y, int ldy, int N, int Hi, int Wi, int Ho, int Wo,
                        int C, float scale_h, float scale_w, float* sums, int sums_ld, void* stream);
/* strided channel-slice copy / add:  dst[:, 0:C] (op)= src[:, 0:C] */
int ydl_copy2d(int dtype, const void* src, int lds, void* dst, int ldd, int64_t npix, int C, int accumulate,
               void* stream);
/* layout/dtype conversion at the model edge: NCHW f32 <-> NHWC compute dtype (channels padded with zeros) */
int ydl_nchw_to_nhwc(int dtype, const float* src, void* dst, int ldd, int N, int C, int H, int W, void* stream);
/* space-to-depth edge conversion for a stem conv with k % s == 0 and p % s == 0 (seg_diceloss_yolov5.py backbone layer 0:
 * Conv(3, 64, 6, 2, 2)):  conv(k,s,p) on (H,W,C) == conv(k/s,1,p/s) on (H/s, W/s, s*s*C), channel (dy*s+dx)*C + c.
 * ydl_weight_prep_s2d / ydl_wgrad_unpack_s2d apply the same index map to the KRSC weight and its gradient. */
int ydl_nchw_to_s2d(int dtype, const float* src, void* dst, int ldd, int N, int C, int H, int W, int s, void* stream);
int ydl_weight_prep_s2d(int dtype, const float* master, void* w2, int Cout, int k, int s, int C, void* stream);
int ydl_wgrad_unpack_s2d(const float* dw2, float* grad, int Cout, int k, int s, int C, int accumulate, void* stream);
int ydl_nhwc_to_nchw(int dtype, const void* src, int lds, float* dst, int N, int C, int H, int W, int accumulate,
                     void* stream);
/* generic elementwise on NHWC slices: out = a * b_bcast ... used by GAM (x * gate[n,c]) */
int ydl_scale_channels(int dtype, const void* x, int ldx, const float* gate /*[N][C]*/, void* y, int ldy,
                       int N, int64_t hw, int C, void* stream);

/* GAM (unet-lite/yolo9-seg/seg_diceloss_yolov9.py:475-510): global average / max pooling to 1x1 (argmax = pixel index of the
 * first maximum, int32 [N][round_up(C,8)]), sigmoid gate of two pooled branches, per-(n,c) dot product */
int ydl_global_pool_fwd(int dtype, const void* x, int ldx, void* avg, int lda, void* mx, int ldm, int32_t* argmax,
                        int N, int64_t HW, int C, void* stream);
int ydl_global_pool_bwd(int dtype, const void* davg, int lda, const void* dmx, int ldm, const int32_t* argmax,
                        void* dx, int lddx, int accumulate, int N, int64_t HW, int C, void* stream);
int ydl_gate_fwd(int dtype, const void* a, int lda, const void* b, int ldb, float* gate, int N, int C, void* stream);
int ydl_gate_bwd(int dtype, const float* gate, const float* dgate, void* da, int lda, int acc_a, void* db, int ldb,
                 int acc_b, int N, int C, void* stream);
int ydl_channel_dot(int dtype, const void* a, int lda, const void* b, int ldb, float* out, int N, int64_t HW, int C,
                    void* stream);

/* ---- softmax over channels (the yaml models end in nn.Softmax(1)) ----------------------------------- */
/* x: NHWC compute dtype (C<=32), stored at (H, W); p: f32 of logical size (N, C, H*rep_h, W*rep_w) with element
 * strides (sn, sc, sh, sw) — NCHW or NHWC.  rep_* > 1 fuses a nearest up-sampling of the probabilities. */
int ydl_softmax_fwd(int dtype, const void* x, int ldx, float* p, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                    int N, int H, int W, int C, int rep_h, int rep_w, void* stream);
/* dx = p * (dps - sum_c p*dps), dps = dp summed over the rep_h x rep_w replicas of each stored pixel */
int ydl_softmax_bwd(int dtype, const float* p, const float* dp, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                    void* dx, int lddx, int N, int H, int W, int C, int rep_h, int rep_w, void* stream);

/* ---- CE + 0.5*(Dice|Jaccard) loss ------------------------------------------------------------------ */
/* ws layout (f32): see ydl_seg_loss_ws_floats.  losses[0..2] = total, ce, overlap-loss (device, no sync). */
int64_t ydl_seg_loss_ws_floats(int N, int C);
int ydl_seg_loss_fwd(const float* pred, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                     const int64_t* target, int Ht, int Wt, const float* class_weights /* NULL = ones */,
                     int kind, float label_smoothing, float eps, int N, int C, int H, int W,
                     float* ws, float* losses, void* stream);
/* dpred = dloss * dL/dpred, using the sums left in ws by the forward */
int ydl_seg_loss_bwd(const float* pred, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                     const int64_t* target, int Ht, int Wt, const float* class_weights,
                     int kind, float label_smoothing, float eps, int N, int C, int H, int W,
                     const float* ws, const float* dloss /* device scalar, NULL = 1 */, float* dpred, void* stream);

/* Replicated-prediction variant of the two calls above: the (N, C, H*rep_h, W*rep_w) prediction is the exact nearest
 * replication of `plow` (N, C, H, W) — the lazily evaluated `Upsample(nearest) -> Conv 1x1 -> Softmax` tail of the yaml
 * models (seg_diceloss_yolov5.py:588-614 builds it).  target is (N, H*rep_h, W*rep_w) int64.  Same ws layout, same
 * `losses`; the backward returns dlow[n,c,h,w] = sum over the rep_h*rep_w replicas of d loss / d pred. */
int ydl_seg_loss_rep_fwd(const float* plow, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                         const int64_t* target, const float* class_weights, int kind, float label_smoothing, float eps,
                         int N, int C, int H, int W, int rep_h, int rep_w, float* ws, float* losses, void* stream);
int ydl_seg_loss_rep_bwd(const float* plow, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                         const int64_t* target, const float* class_weights, int kind, float label_smoothing, float eps,
                         int N, int C, int H, int W, int rep_h, int rep_w, const float* ws,
                         const float* dloss /* device scalar, NULL = 1 */, float* dlow, void* stream);

/* ---- optimizer: SGD(nesterov) + EMA on flat arenas --------------------------------------------------- */
/* params/grads/momentum are flat f32 arenas laid out [decay group | no-decay group]; ema spans
 * n_params + n_buffers (BN running stats follow the params in both `params` and `ema`).
 * first_step != 0: momentum buffer is initialised with the gradient (torch.optim.SGD semantics). */
int ydl_sgd_ema_step(float* params, const float* grads, float* momentum, float* ema,
                     int64_t n_decay, int64_t n_params, int64_t n_total,
                     float lr_decay_group, float lr_nodecay_group, float mom, float weight_decay, float grad_scale,
                     int first_step, float ema_decay /* <0: skip EMA */, void* stream);

/* graph-capturable form: hyper_dev = device float[7] {lr_weights, lr_bn, lr_bias, momentum, weight_decay, grad_scale,
 * ema_decay}; lr_index selects the learning rate of this run (0 weights, 1 BN weights, 2 biases) */
int ydl_sgd_ema_step_dev(float* params, const float* grads, float* momentum, float* ema,
                         int64_t n_decay, int64_t n_params, int64_t n_total, const float* hyper_dev,
                         int lr_index, int use_weight_decay, int first_step, int use_ema, void* stream);

/* every run of one step in one launch: runs_dev = device int64[nruns][6] rows {offset (elements into all four arenas), n_decay,
 * n_params, n_total, lr_index, flags: bit 0 weight decay, bit 1 first step}, each row with the meaning of the arguments of
 * ydl_sgd_ema_step_dev applied at `offset`; max_run = the largest n_total (grid sizing). */
int ydl_sgd_ema_step_multi(float* params, const float* grads, float* momentum, float* ema, const int64_t* runs_dev,
                           int nruns, int64_t max_run, const float* hyper_dev, int use_ema, void* stream);

/* ---- evaluation: argmax + confusion matrix (val_diceloss.py:37-75) ---------------------------------- */
int ydl_confusion_matrix(const float* pred, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                         const int64_t* target, int N, int C, int H, int W, int ignore_index,
                         int64_t* matrix /* [C][C], accumulated */, void* stream);

/* ---- DCNv3 (models/ops_dcnv3/src/cuda/dcnv3_cuda.h:15-31) -------------------------------------------------------
 * Forward: the reference's argument order (tensors, kernel/stride/pad/dilation, group, group_channels, offset_scale),
 * followed by the sizes the reference reads off its tensors.  Backward: the same, except that grad_output sits with the
 * other tensors (the reference passes it after offset_scale, dcnv3_cuda.h:24-31) and the three gradients are caller-owned
 * outputs instead of a returned vector.  `im2col_step` has no counterpart (no batch chunking).  dtype also accepts
 * YDL_F16 here (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF, dcnv3_cuda.cu:69,147). */
int ydl_dcnv3_fwd(int dtype, const void* input, const void* offset, const void* mask, void* output,
                  int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w,
                  int dilation_h, int dilation_w, int group, int group_channels, float offset_scale,
                  int N, int H_in, int W_in, int H_out, int W_out, void* stream);
/* Border rule of the sampling positions (process-wide, read at launch time), the one case where the reference has two answers:
 *   0 (default)  a position exactly at -1 is inside, ">= -1": dcnv3_core_pytorch, functions/dcnv3_func.py:148-189 (F.grid_sample,
 *                padding zeros) — the form of the op the oracle and every golden vector of this repo come from;
 *   1            "> -1" as the CUDA op tests it, dcnv3_im2col_cuda.cuh:262,334,428: such a point has no value and no gradient.
 * A binding that replaces DCNv3Function's CUDA extension selects 1 once at import time. */
void ydl_dcnv3_set_border_rule(int rule);
int ydl_dcnv3_get_border_rule(void);
/* grad_input must be zeroed by the caller (f32 atomics); grad_offset/grad_mask are fully written. */
int ydl_dcnv3_bwd(int dtype, const void* input, const void* offset, const void* mask, const void* grad_output,
                  float* grad_input, float* grad_offset, float* grad_mask,
                  int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w,
                  int dilation_h, int dilation_w, int group, int group_channels, float offset_scale,
                  int N, int H_in, int W_in, int H_out, int W_out, void* stream);


/* ---- pieces of the DCNv3 module around the sampling op (models/ops_dcnv3/build/.../modules/dcnv3.py:50-136) ------------
 * depth-wise k x k convolution, stride 1, 'same' padding (the `dw_conv = Conv(c, c, k, g=c)` branch, :89);
 * w is the f32 master weight [C][k*k] (= nn.Conv2d(C, C, k, groups=C).weight, shape [C,1,k,k]); k in {1,3,5,7}. */
int ydl_dwconv_fwd(int dtype, const void* x, int ldx, const float* w, void* y, int ldy, int N, int H, int W, int C,
                   int k, int p, void* stream);
int ydl_dwconv_dgrad(int dtype, const void* dy, int lddy, const float* w, void* dx, int lddx, int accumulate, int N, int H, int W,
                     int C, int k, int p, void* stream);
/* dw[C][k*k] += sum over pixels; deterministic (per-block partials in ws, fixed-order merge) */
int64_t ydl_dwconv_wgrad_ws_bytes(int C, int k);
int ydl_dwconv_wgrad(int dtype, const void* x, int ldx, const void* dy, int lddy, float* dw, float* ws, int N, int H, int W, int C,
                     int k, int p, void* stream);
/* BN partial statistics of any NHWC tensor in the format ydl_bn_finalize consumes: nblocks = ceil(npix / block_m) rows of
 * (sum, M2) with block_m = ydl_bn_stats_block_m() */
int ydl_bn_stats_block_m(void);
int64_t ydl_bn_stats_ws_bytes(int64_t npix, int C);
int ydl_bn_stats(int dtype, const void* y, int ldy, float* ws, int64_t npix, int C, void* stream);
/* out[c] (+)= sum over pixels x[p][c]  (bias gradient of the NHWC nn.Linear layers); deterministic */
int64_t ydl_channel_sum_ws_bytes(int C);
int ydl_channel_sum(int dtype, const void* x, int ldx, float* out, float* ws, int64_t npix, int C, int accumulate, void* stream);
/* soft-max over the P = K*K sampling points of each of the G groups (modules/dcnv3.py:122-123); rows of G*P values */
int ydl_group_softmax_fwd(int dtype, const void* x, int ldx, void* y, int ldy, int64_t npix, int G, int P, void* stream);
int ydl_group_softmax_bwd(int dtype, const void* y, int ldy, const void* dy, int lddy, void* dx, int lddx, int accumulate,
                          int64_t npix, int G, int P, void* stream);
/* dst[p][0:C] (+)= (compute dtype) src[p][0:C]; src f32 (the DCNv3 op returns f32 gradients, dcnv3_cuda.cu:126-133) */
int ydl_cast_f32(int dtype, const float* src, int lds, void* dst, int ldd, int64_t npix, int C, int accumulate, void* stream);

/* ---- gradient transport helpers (yolo_dual_amd.parallel: reduce-scatter + all-gather over all peers, bf16 wire) ---------
 * dst[i] = sum over r < nchunks of src[r][i] in that fixed order (src f32 or bf16, dst f32);  dst (+)= (f32) src */
int ydl_reduce_chunks(int dtype, const void* src, float* dst, int64_t n, int nchunks, void* stream);
int ydl_cast_to_f32(int dtype, const void* src, float* dst, int64_t n, int accumulate, void* stream);

/* ---- GPU half of the per-sample input preparation -------------------------------------------------------
 * JSONSegmentDataset._resize_and_pad + the format conversion of __getitem__ (unet-lite/yolo5-seg/seg_diceloss_yolov5.py:
 * 309-349): img.resize((new_w,new_h), Image.BILINEAR) pasted on a (128,128,128) canvas, /255, HWC -> CHW float32.
 * src: uint8 [h][w][3] on the device; dst: float [3][S][S]; tmp: uint8 [h][new_w][3] scratch (may be NULL when new_w == w).
 * The arithmetic is Pillow's ImagingResample for 8-bit images (horizontal pass, 8-bit intermediate, vertical pass; 22-bit
 * fixed-point coefficients): the tables are built on the host in double exactly like Pillow's precompute_coeffs /
 * normalize_coeffs_8bpc (yolo_dual_amd/data.py) — bounds int [out][2] = (first source index, tap count), coef int
 * [out][ksize].  The kernels trust the tables (first + count <= source size).  fill = canvas grey level (128). */
int ydl_letterbox_image(const void* src, int h, int w, void* tmp, float* dst, int S, int new_w, int new_h,
                        int pad_left, int pad_top, const int* xbounds, const int* xcoef, int xksize,
                        const int* ybounds, const int* ycoef, int yksize, int fill, void* stream);
/* mask.resize((new_w,new_h), Image.NEAREST) pasted on a 0 canvas, np.clip(., 0, clip_max) (:303), int64 (:315).
 * src: uint8 [h][w]; dst: int64 [S][S]; xtab int[new_w] / ytab int[new_h]: Pillow's ImagingScaleAffine source indices
 * (sequential double additions, built on the host). */
int ydl_letterbox_mask(const void* src, int h, int w, int64_t* dst, int S, int new_w, int new_h, int pad_left,
                       int pad_top, const int* xtab, const int* ytab, int clip_max, void* stream);

/* ---- zero fills (the taped region issues no ATen kernel: buffers that need defined contents are cleared through these) ----
 * ydl_fill_zero: `bytes` bytes at dst (hipMemsetAsync);  ydl_zero2d: dst[p][0:C] = 0 for npix rows of pixel stride ldd */
int ydl_fill_zero(void* dst, int64_t bytes, void* stream);
int ydl_zero2d(int dtype, void* dst, int ldd, int64_t npix, int C, void* stream);

/* ---- launch-list replay: the training step without its host wall -------------------------------------------------------
 * The reference's hot loop (seg_diceloss_yolov5.py:1084-1103) issues the same sequence of device work every step.  The host
 * side records ONE step — every entry-point call above with its arguments, the stream it went to, and the cross-stream
 * event edges — into a ydl_replay and re-issues it with one call per step: no Python, no ctypes marshalling per launch, the
 * same kernels on the same streams with the same dependencies as the eager step (so the eager step's overlap survives).
 * Preconditions (the caller's job, yolo_dual_amd/replay.py): every buffer a recorded call names must stay valid and at the same
 * address for the life of the ydl_replay (a private allocator pool), and every device operation of the step must be in the
 * list.  Streams are named by slot: the table passed to ydl_replay_run maps slot -> hipStream_t. */
typedef struct ydl_replay ydl_replay;
ydl_replay* ydl_replay_create(void);
void ydl_replay_destroy(ydl_replay* r);
/* number / names of the recordable entry points (index = `fn` below); generated from this header (tools/gen_replay.py) */
int ydl_replay_fn_count(void);
const char* ydl_replay_fn_name(int fn);
/* append one call: args = the entry point's parameters before `stream`, one 8-byte slot each (integers and pointers as
 * int64, floats as the bits of a double; a ydl_conv_geom* / ydl_bnred* slot holds a HOST pointer whose struct is copied) */
int ydl_replay_add_call(ydl_replay* r, int fn, const int64_t* args, int nargs, int stream_slot);
/* cross-stream edge: hipEventRecord(event[id], stream[slot]) / hipStreamWaitEvent(stream[slot], event[id]) */
int ydl_replay_add_event_record(ydl_replay* r, int event_id, int stream_slot);
int ydl_replay_add_event_wait(ydl_replay* r, int event_id, int stream_slot);
int ydl_replay_size(const ydl_replay* r);
/* re-issue operations [first, last) in order; h_streams: host array of nstreams hipStream_t.  Stops at the first failing call
 * (its status is returned, ydl_last_error() says which operation). */
int ydl_replay_run(ydl_replay* r, int first, int last, void* const* h_streams, int nstreams);

#ifdef __cplusplus
}
#endif
#endif /* YDL_H_ */
