// DCNv3 (grouped, mask-modulated deformable sampling, NHWC) forward / backward for gfx950.
// Behaviour follows models/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:216-275 (forward) and :82-147 (gradients);
// the decomposition is ours: a (pixel, group) item is owned by a power-of-two lane segment of one wavefront
// (lanes = group channels, so every NHWC access is a contiguous run), and the grad_offset / grad_mask sums over the
// channels are wavefront-shuffle butterflies instead of the reference's shared-memory trees.
// One measure-zero case has TWO conventions in the reference, selectable here (ydl_dcnv3_set_border_rule):
//   rule 0 (default) — the pure-PyTorch core, functions/dcnv3_func.py:148-189 (the only form of the op the reference can run without
//     its missing binding, hence the oracle): a sampling point EXACTLY at -1 is inside (">= -1").  Its value is zero either way, but
//     its offset gradient is the one-sided slope towards pixel 0, as grid_sample's backward gives;
//   rule 1 — the CUDA op, dcnv3_im2col_cuda.cuh:262,334,428: "loc > -1" — a point exactly at -1 is outside: no value, no gradient.
// With the module's own initialisation (offset weights zero, pad 1) every border tap sits exactly at -1, so the two rules give
// different offset-bias gradients on the first step.
#include "common.h"
#include <stdlib.h>

// fp16 storage (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF, dcnv3_cuda.cu:69,147); only this op takes it
template <> struct ET<_Float16> {
    static constexpr int V = 8;
    __device__ static __forceinline__ float ld(const _Float16* p) { return (float)*p; }
    __device__ static __forceinline__ void st(_Float16* p, float v) { *p = (_Float16)v; }
};
template <> __device__ __forceinline__ void unpack16<_Float16>(const uint4& u, float* f) {
    const _Float16* h = (const _Float16*)&u;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (float)h[e];
}
template <> __device__ __forceinline__ uint4 pack16<_Float16>(const float* f) {
    uint4 u;
    _Float16* h = (_Float16*)&u;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (_Float16)f[e];
    return u;
}

struct DcnArgs {
    const void* in; const void* off; const void* msk; void* out;        // fwd
    const void* gout; float* gin; float* goff; float* gmsk;             // bwd
    int kh, kw, sh, sw, ph, pw, dh, dw, G, Gc;
    float scale;
    int N, H, W, Ho, Wo;
    int seg;          // lanes per item (power of two, <= 64)
    long long items;  // N*Ho*Wo*G
    int strict;       // border rule: 0 ">= -1" (PyTorch core), 1 "> -1" (.cuh)
    int dbg;          // timing experiments only (YDL_DCN_DBG): bit 0 no flush atomics, bit 1 no MFMA phase, bit 2 no direct atomics,
                      //                                          bit 3 no gathers (corner values zero)
};

static int g_dcn_border_rule = 0;
static int g_dcn_win = 1;          // ydl_debug_set key 13 (YDL_DCN_NOWIN=1 at start-up switches the window backward off)
static int g_dcn_tile = 1;         // ydl_debug_set key 16: 1 (default) tile backward (S x grad_output on the MFMA) where it applies and the
                                   // map has two rounds of tiles, 2 = at any size (tests), 0 off
extern "C" void ydl_dcnv3_set_border_rule(int rule) { g_dcn_border_rule = rule ? 1 : 0; }
extern "C" int ydl_dcnv3_get_border_rule(void) { return g_dcn_border_rule; }
void ydl_dcn_debug_set(int key, int val) { if (key == 13) g_dcn_win = val; if (key == 16) g_dcn_tile = val; }

// inside test of a sampling position: lower edge by the selected rule, upper edge exclusive in both conventions
__device__ __forceinline__ bool dcn_inside(float lh_, float lw_, int H, int W, int strict) {
    const bool lo = strict ? (lh_ > -1.f && lw_ > -1.f) : (lh_ >= -1.f && lw_ >= -1.f);
    return lo && lh_ < (float)H && lw_ < (float)W;
}

__device__ __forceinline__ float seg_sum(float v, int seg) {
    for (int o = seg >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void dcnv3_kernel(const DcnArgs a) {
    const int lane = threadIdx.x & 63;
    const int ipw = 64 / a.seg;                          // items per wave
    const int sub = lane / a.seg, cl = lane % a.seg;
    const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int P = a.kh * a.kw;
    const int C = a.G * a.Gc;
    const T* in = (const T*)a.in;
    const T* off = (const T*)a.off;
    const T* msk = (const T*)a.msk;
    const long long rounds = (a.items + ipw - 1) / ipw;
    for (long long rd = wave_id; rd < rounds; rd += nwaves) {
        long long item = rd * ipw + sub;
        bool live = item < a.items;
        long long it = live ? item : 0;
        int g = (int)(it % a.G);
        long long pix = it / a.G;
        int wo = (int)(pix % a.Wo);
        long long t2 = pix / a.Wo;
        int ho = (int)(t2 % a.Ho);
        int n = (int)(t2 / a.Ho);
        const int p0w = ((a.dw * (a.kw - 1)) >> 1) - a.pw + wo * a.sw;
        const int p0h = ((a.dh * (a.kh - 1)) >> 1) - a.ph + ho * a.sh;
        const float p0w_ = (float)p0w - (float)((a.dw * (a.kw - 1)) >> 1) * a.scale;
        const float p0h_ = (float)p0h - (float)((a.dh * (a.kh - 1)) >> 1) * a.scale;
        const T* offp = off + (size_t)pix * a.G * P * 2 + (size_t)g * P * 2;
        const T* mskp = msk + (size_t)pix * a.G * P + (size_t)g * P;
        const T* imb = in + (size_t)n * a.H * a.W * C + (size_t)g * a.Gc;
        for (int c0 = 0; c0 < a.Gc; c0 += a.seg) {
            const int c = c0 + cl;
            const bool act = live && c < a.Gc;
            float col = 0.f;
            float go = 0.f;
            if (BWD && act) go = ET<T>::ld((const T*)a.gout + (size_t)pix * C + g * a.Gc + c);
            int k = 0;
            for (int i = 0; i < a.kw; ++i)
                for (int j = 0; j < a.kh; ++j, ++k) {
                    float ow = live ? ET<T>::ld(offp + 2 * k) : 0.f, oh = live ? ET<T>::ld(offp + 2 * k + 1) : 0.f;
                    float mk = live ? ET<T>::ld(mskp + k) : 0.f;
                    float lw_ = p0w_ + ((float)(i * a.dw) + ow) * a.scale;
                    float lh_ = p0h_ + ((float)(j * a.dh) + oh) * a.scale;
                    float gmask = 0.f, goffw = 0.f, goffh = 0.f;
                    if (dcn_inside(lh_, lw_, a.H, a.W, a.strict)) {
                        int hl = (int)floorf(lh_), wl = (int)floorf(lw_);
                        int hh_ = hl + 1, wh_ = wl + 1;
                        float lh = lh_ - (float)hl, lw = lw_ - (float)wl;
                        float hh = 1.f - lh, hw = 1.f - lw;
                        bool b1 = hl >= 0 && wl >= 0, b2 = hl >= 0 && wh_ <= a.W - 1;
                        bool b3 = hh_ <= a.H - 1 && wl >= 0, b4 = hh_ <= a.H - 1 && wh_ <= a.W - 1;
                        size_t o1 = ((size_t)hl * a.W + wl) * C + c, o2 = ((size_t)hl * a.W + wh_) * C + c;
                        size_t o3 = ((size_t)hh_ * a.W + wl) * C + c, o4 = ((size_t)hh_ * a.W + wh_) * C + c;
                        float v1 = (act && b1) ? ET<T>::ld(imb + o1) : 0.f;
                        float v2 = (act && b2) ? ET<T>::ld(imb + o2) : 0.f;
                        float v3 = (act && b3) ? ET<T>::ld(imb + o3) : 0.f;
                        float v4 = (act && b4) ? ET<T>::ld(imb + o4) : 0.f;
                        float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                        float val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
                        if (!BWD) {
                            col += val * mk;
                        } else if (act) {
                            float tg = go * mk;
                            float* gib = a.gin + (size_t)n * a.H * a.W * C + (size_t)g * a.Gc;
                            // a corner with bilinear weight 0 adds nothing: skip its atomic.  (Integer-aligned sampling positions — the
                            // module starts with zero offsets, modules/dcnv3.py:101-107 — then touch one address per point, not four.)
                            if (b1 && w1 != 0.f) atomicAdd(gib + o1, w1 * tg);
                            if (b2 && w2 != 0.f) atomicAdd(gib + o2, w2 * tg);
                            if (b3 && w3 != 0.f) atomicAdd(gib + o3, w3 * tg);
                            if (b4 && w4 != 0.f) atomicAdd(gib + o4, w4 * tg);
                            float ghw = -hw * v1 - lw * v2 + hw * v3 + lw * v4;   // d val / d h
                            float gww = -hh * v1 + hh * v2 - lh * v3 + lh * v4;   // d val / d w
                            gmask = go * val;
                            goffw = a.scale * gww * tg;
                            goffh = a.scale * ghw * tg;
                        }
                    }
                    if (BWD) {
                        // every lane of the wave takes part in the butterflies (uniform control flow)
                        gmask = seg_sum(gmask, a.seg);
                        goffw = seg_sum(goffw, a.seg);
                        goffh = seg_sum(goffh, a.seg);
                        if (live && cl == 0) {
                            float* gof = a.goff + (size_t)pix * a.G * P * 2 + (size_t)g * P * 2 + 2 * k;
                            float* gmk = a.gmsk + (size_t)pix * a.G * P + (size_t)g * P + k;
                            if (c0 == 0) { gof[0] = goffw; gof[1] = goffh; gmk[0] = gmask; }
                            else { gof[0] += goffw; gof[1] += goffh; gmk[0] += gmask; }
                        }
                    }
                }
            if (!BWD && act) ET<T>::st((T*)a.out + (size_t)pix * C + g * a.Gc + c, col);
        }
    }
}


// ------------------------------------------------------------------------------------------------------
// Backward for 3 x 3 kernels with >= 33 channels per group (a whole wavefront per (pixel, group) item, lane = channel), round 4.
// The plain kernel above issues one f32 atomic wave-instruction per (point, bilinear corner): 36 per item and 64-channel pass, and the
// op sits at the ~1.3 TB/s device-scope atomic rate.  With a whole wave on one item every sampling position is WAVE-UNIFORM, so the
// corner gradients can be merged in registers first: a 5 x 5 window of dx cells anchored one cell up-left of the un-shifted top-left
// tap (offsets below one pixel keep all 36 corners inside it) is a private float[25] per lane, indexed with a uniform index
// (s_set_gpr_idx, no scratch, no LDS); at the end of the pass the touched cells go out with ONE atomic each — 16 for offsets near
// zero, at most 25 — and corners that fall outside the window (large offsets) keep their direct atomic.  Same products as the plain
// kernel, merged before instead of inside the memory system; grad_offset / grad_mask sums use DPP row adds + row broadcasts
// (v_add_f32_dpp) instead of ds_bpermute butterflies.  Reference arithmetic: dcnv3_im2col_cuda.cuh:82-147.
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dcnv3_bwd_win_kernel(const DcnArgs a) {
    const int lane = threadIdx.x & 63;
    const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    constexpr int P = 9;
    const int C = a.G * a.Gc;
    const T* in = (const T*)a.in;
    const T* off = (const T*)a.off;
    const T* msk = (const T*)a.msk;
    for (long long it = wave_id; it < a.items; it += nwaves) {
        // (pixel, group) of this wave: uniform — say so, the window index below must be a scalar
        const int g = __builtin_amdgcn_readfirstlane((int)(it % a.G));
        const long long pix = it / a.G;
        const int wo = __builtin_amdgcn_readfirstlane((int)(pix % a.Wo));
        const long long t2 = pix / a.Wo;
        const int ho = __builtin_amdgcn_readfirstlane((int)(t2 % a.Ho));
        const int n = __builtin_amdgcn_readfirstlane((int)(t2 / a.Ho));
        const int p0w = ((a.dw * 2) >> 1) - a.pw + wo * a.sw;
        const int p0h = ((a.dh * 2) >> 1) - a.ph + ho * a.sh;
        const float p0w_ = (float)p0w - (float)((a.dw * 2) >> 1) * a.scale;
        const float p0h_ = (float)p0h - (float)((a.dh * 2) >> 1) * a.scale;
        const int hb = (int)floorf(p0h_) - 1, wb = (int)floorf(p0w_) - 1;          // window anchor (cell (0, 0))
        const T* offp = off + (size_t)pix * a.G * P * 2 + (size_t)g * P * 2;
        const T* mskp = msk + (size_t)pix * a.G * P + (size_t)g * P;
        const T* imb = in + (size_t)n * a.H * a.W * C + (size_t)g * a.Gc;
        float* gib = a.gin + (size_t)n * a.H * a.W * C + (size_t)g * a.Gc;
        for (int c0 = 0; c0 < a.Gc; c0 += 64) {
            const int c = c0 + lane;
            const bool act = c < a.Gc;
            const float go = act ? ET<T>::ld((const T*)a.gout + (size_t)pix * C + g * a.Gc + c) : 0.f;
            static_assert(25 <= 32, "one bit of `touched` per window cell");
            float win[25];
#pragma unroll
            for (int q = 0; q < 25; ++q) win[q] = 0.f;
            unsigned touched = 0;                                   // uniform: bit q = cell q received a contribution
            int k = 0;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j, ++k) {
                    const float ow = ET<T>::ld(offp + 2 * k), oh = ET<T>::ld(offp + 2 * k + 1), mk = ET<T>::ld(mskp + k);
                    const float lw_ = p0w_ + ((float)(i * a.dw) + ow) * a.scale;
                    const float lh_ = p0h_ + ((float)(j * a.dh) + oh) * a.scale;
                    float gmask = 0.f, goffw = 0.f, goffh = 0.f;
                    if (dcn_inside(lh_, lw_, a.H, a.W, a.strict)) {          // (uniform)
                        const int hl = __builtin_amdgcn_readfirstlane((int)floorf(lh_)), wl = __builtin_amdgcn_readfirstlane((int)floorf(lw_));
                        const int hh_ = hl + 1, wh_ = wl + 1;
                        const float lh = lh_ - (float)hl, lw = lw_ - (float)wl;
                        const float hh = 1.f - lh, hw = 1.f - lw;
                        const bool b1 = hl >= 0 && wl >= 0, b2 = hl >= 0 && wh_ <= a.W - 1;
                        const bool b3 = hh_ <= a.H - 1 && wl >= 0, b4 = hh_ <= a.H - 1 && wh_ <= a.W - 1;
                        const size_t o1 = ((size_t)hl * a.W + wl) * C + c, o2 = ((size_t)hl * a.W + wh_) * C + c;
                        const size_t o3 = ((size_t)hh_ * a.W + wl) * C + c, o4 = ((size_t)hh_ * a.W + wh_) * C + c;
                        const float v1 = (act && b1) ? ET<T>::ld(imb + o1) : 0.f;
                        const float v2 = (act && b2) ? ET<T>::ld(imb + o2) : 0.f;
                        const float v3 = (act && b3) ? ET<T>::ld(imb + o3) : 0.f;
                        const float v4 = (act && b4) ? ET<T>::ld(imb + o4) : 0.f;
                        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                        const float val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
                        const float tg = go * mk;
                        const int r = hl - hb, cc = wl - wb;                      // window cell of corner 1 (uniform)
                        auto corner = [&](bool b, float w, int rr, int cq, size_t o) {
                            if (b && w != 0.f) {                                  // a corner with bilinear weight 0 adds nothing
                                if ((unsigned)rr < 5u && (unsigned)cq < 5u) {
                                    const int q = rr * 5 + cq;
                                    win[q] += w * tg;
                                    touched |= 1u << q;
                                } else if (act) {
                                    atomicAdd(gib + o, w * tg);
                                }
                            }
                        };
                        corner(b1, w1, r, cc, o1);
                        corner(b2, w2, r, cc + 1, o2);
                        corner(b3, w3, r + 1, cc, o3);
                        corner(b4, w4, r + 1, cc + 1, o4);
                        const float ghw = -hw * v1 - lw * v2 + hw * v3 + lw * v4;   // d val / d h
                        const float gww = -hh * v1 + hh * v2 - lh * v3 + lh * v4;   // d val / d w
                        gmask = go * val;
                        goffw = a.scale * gww * tg;
                        goffh = a.scale * ghw * tg;
                    }
                    gmask = wave_total63(gmask);
                    goffw = wave_total63(goffw);
                    goffh = wave_total63(goffh);
                    if (lane == 63) {
                        float* gof = a.goff + (size_t)pix * a.G * P * 2 + (size_t)g * P * 2 + 2 * k;
                        float* gmk = a.gmsk + (size_t)pix * a.G * P + (size_t)g * P + k;
                        if (c0 == 0) { gof[0] = goffw; gof[1] = goffh; gmk[0] = gmask; }
                        else { gof[0] += goffw; gof[1] += goffh; gmk[0] += gmask; }
                    }
                }
            touched = __builtin_amdgcn_readfirstlane(touched);
#pragma unroll
            for (int q = 0; q < 25; ++q)
                if ((touched >> q) & 1u) {
                    // (uniform) a bit is only set for a corner that passed its own bounds test (b1..b4), so the cell lies inside the
                    // image; the test below costs two scalar compares and makes that a property of THIS loop, not of the code above
                    // (the withdrawn 7 x 7 form kept 49 cells behind this 32-bit mask: `1u << q` wrapped for q >= 32, marked cells that
                    // no corner had vouched for, and border pixels then flushed to rows in front of the tensor — the round-4 GPU fault)
                    const int fh = hb + q / 5, fw = wb + q % 5;
                    if (act && (unsigned)fh < (unsigned)a.H && (unsigned)fw < (unsigned)a.W)
                        atomicAdd(gib + ((size_t)fh * a.W + fw) * C + c, win[q]);
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Backward for 3 x 3 / stride 1 / dilation 1 / offset_scale 1, group widths that are multiples of 64, round 5: the grad_input scatter as a MATRIX
// PRODUCT.  What every corner adds is (bilinear weight x mask) — one scalar per (output pixel, corner), the same for all channels —
// times grad_output[pixel][c]:      grad_input[cell][c] = sum_pixels S[cell][pixel] * grad_output[pixel][c].
// A CTA owns an 8 x 8 tile of output pixels of one (image, group) and the 18 x 18 window of input cells that sampling offsets
// below 4 pixels can reach (T_R = 4: 324 cells).  Phase 1, a wave per pixel as in the kernels above (lanes = channels: gathers of the
// four corners, grad_offset / grad_mask by DPP wave totals): the corner scalars do NOT go to memory — the wave keeps its pixel's
// column of S in registers (cell q in lane q & 63, slot q >> 6; a uniform cell index, one lane adds) and writes it once, together
// with its grad_output row (f32), to LDS: no atomics, no zero fill, every (pixel, cell) is written.  Phase 2: S x GO on the f32 MFMA
// (v_mfma_f32_16x16x4_f32: exact f32 products, 21 x 4 tiles x 16 k-steps per CTA and 64-channel chunk),
// and the in-image, non-zero cells leave the accumulator registers as they stand (an atomic instruction = 4 cells x 16 channels: four
// 64-byte pieces, the four memory-side requests a 256-byte row would make too): 83 KB of atomics per 64 pixels and 64 channels
// instead of 590 KB, whatever the offsets are.  Groups wider than 64 channels run chunk by chunk with the SAME S (built once).
// Corners beyond the window (|offset| >= 4 px) keep their direct atomics.
// Reference arithmetic: dcnv3_im2col_cuda.cuh:82-147 (same products; the sum over pixels is formed in the MFMA's order).
// ------------------------------------------------------------------------------------------------------
#define T_R 4
#define T_WIN (10 + 2 * T_R)                 // window edge: taps th-p .. th+9-p, +-R, +1 for the upper bilinear corner
#define T_CELLS (T_WIN * T_WIN)              // 324
#define T_CPAD ((T_CELLS + 15) / 16 * 16)    // 336: whole 16-row MFMA tiles
#define T_SLOTS ((T_CPAD + 63) / 64)         // 6 column registers per lane
#define T_GOLD 80                            // GO row stride (floats): 16 banks per row => conflict-free 16x16x4 B-operand reads
typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define T_REC 8                               // dwords per (pixel, point) record
template <typename T>
__global__ __launch_bounds__(1024) void dcnv3_bwd_tile_kernel(const DcnArgs a, int tiles_w, int tiles_hw) {
    extern __shared__ __attribute__((aligned(16))) float slds[];
    float* const sS = slds;                              // [64 pixels][T_CPAD cells]  (S transposed: a pixel's column is contiguous)
    float* const sGO = slds + 64 * T_CPAD;               // [64 pixels][T_GOLD]: grad_output of the current 64-channel chunk
    int* const sRec = (int*)(sGO + 64 * T_GOLD);         // [64 pixels][9 points][T_REC]: everything about a sampling point that does not
                                                         // depend on the channel, computed ONCE by one lane (pre-pass below)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // 16 waves, 4 pixels each
    constexpr int P = 9;
    const int C = a.G * a.Gc;
    int b = blockIdx.x;
    const int g = b % a.G; b /= a.G;
    const int tl = b % tiles_hw;
    const int n = b / tiles_hw;
    const int th = (tl / tiles_w) * 8, tw = (tl % tiles_w) * 8;
    const int wh0 = th - a.ph - T_R, ww0 = tw - a.pw - T_R;                 // input cell of window cell (0, 0)
    const T* off = (const T*)a.off;
    const T* msk = (const T*)a.msk;
    const int lrow = lane & 15, lgrp = lane >> 4;
    // ---- pre-pass: a lane per (pixel, point).  A wave per pixel did this arithmetic 64 times over (position, floor, four bounds tests,
    // four 64-bit offsets per point: the kernel was bound by exactly that — with every atomic, gather and MFMA switched off it kept
    // 75 % of its time).  Record: [0] element offset of corner 1 in the image plane ((hl W + wl) C, may be negative), [1] flags: bit 0
    // inside, bits 1-4 corner usable (in the image, non-zero weight), bits 5-8 corner inside the window, [2] window cell of corner 1,
    // [3] lh, [4] lw, [5] mask
    for (int e = threadIdx.x; e < 64 * P; e += 1024) {
        const int item = e / P, k = e - item * P;
        const int ii = k / 3, jj = k - ii * 3;
        const int ho = th + (item >> 3), wo = tw + (item & 7);
        int base = 0, flags = 0, q1 = 0;
        float lh = 0.f, lw = 0.f, mk = 0.f;
        if (ho < a.Ho && wo < a.Wo) {
            const long long pix = ((long long)n * a.Ho + ho) * a.Wo + wo;
            const T* offp = off + (size_t)pix * a.G * P * 2 + (size_t)g * P * 2;
            const float ow = ET<T>::ld(offp + 2 * k), oh = ET<T>::ld(offp + 2 * k + 1);
            mk = ET<T>::ld(msk + (size_t)pix * a.G * P + (size_t)g * P + k);
            const float lw_ = ((float)(1 - a.pw + wo) - 1.f) + ((float)ii + ow);                  // dilation 1, offset_scale 1
            const float lh_ = ((float)(1 - a.ph + ho) - 1.f) + ((float)jj + oh);
            if (dcn_inside(lh_, lw_, a.H, a.W, a.strict)) {
                const int hl = (int)floorf(lh_), wl = (int)floorf(lw_);
                lh = lh_ - (float)hl; lw = lw_ - (float)wl;
                const float hh = 1.f - lh, hw = 1.f - lw;
                const bool b1 = hl >= 0 && wl >= 0, b2 = hl >= 0 && wl + 1 <= a.W - 1;
                const bool b3 = hl + 1 <= a.H - 1 && wl >= 0, b4 = hl + 1 <= a.H - 1 && wl + 1 <= a.W - 1;
                const int r = hl - wh0, cc = wl - ww0;
                const bool i1 = (unsigned)r < (unsigned)T_WIN && (unsigned)cc < (unsigned)T_WIN;
                const bool i2 = (unsigned)r < (unsigned)T_WIN && (unsigned)(cc + 1) < (unsigned)T_WIN;
                const bool i3 = (unsigned)(r + 1) < (unsigned)T_WIN && (unsigned)cc < (unsigned)T_WIN;
                const bool i4 = (unsigned)(r + 1) < (unsigned)T_WIN && (unsigned)(cc + 1) < (unsigned)T_WIN;
                flags = 1 | ((b1 && hh * hw != 0.f) ? 2 : 0) | ((b2 && hh * lw != 0.f) ? 4 : 0) | ((b3 && lh * hw != 0.f) ? 8 : 0) |
                        ((b4 && lh * lw != 0.f) ? 16 : 0) | (i1 ? 32 : 0) | (i2 ? 64 : 0) | (i3 ? 128 : 0) | (i4 ? 256 : 0) |
                        (b1 ? 512 : 0) | (b2 ? 1024 : 0) | (b3 ? 2048 : 0) | (b4 ? 4096 : 0);          // bits 9-12: corner in the image
                base = (hl * a.W + wl) * C;                               // (|.| < 2^31: the plane has fewer than 2^31 elements)
                q1 = r * T_WIN + cc;
            }
        }
        int* rec = sRec + e * T_REC;
        rec[0] = base; rec[1] = flags; rec[2] = q1;
        rec[3] = __float_as_int(lh); rec[4] = __float_as_int(lw); rec[5] = __float_as_int(mk);
    }
    __syncthreads();
    // channel chunks of 64 (Gc is a multiple of 64): S is built with the first chunk and serves all of them
    for (int c0 = 0; c0 < a.Gc; c0 += 64) {
        const T* imb = (const T*)a.in + (size_t)n * a.H * a.W * C + (size_t)g * a.Gc + c0 + lane;
        float* gib = a.gin + (size_t)n * a.H * a.W * C + (size_t)g * a.Gc + c0;
        const int rowC = a.W * C;
        for (int i = 0; i < 4; ++i) {
            const int item = wave * 4 + i;
            const int ho = th + (item >> 3), wo = tw + (item & 7);
            float col[T_SLOTS];
#pragma unroll
            for (int sl = 0; sl < T_SLOTS; ++sl) col[sl] = 0.f;
            float go = 0.f;
            float r_off = 0.f, r_msk = 0.f;            // lane 2k / 2k+1: grad_offset (w, h) of point k; lane k: grad_mask of point k
            const bool live = ho < a.Ho && wo < a.Wo;                          // (uniform)
            long long pix = 0;
            if (live) {
                pix = ((long long)n * a.Ho + ho) * a.Wo + wo;
                go = ET<T>::ld((const T*)a.gout + (size_t)pix * C + g * a.Gc + c0 + lane);
                const int* recp = sRec + item * P * T_REC;
#pragma unroll 1
                for (int k = 0; k < P; ++k, recp += T_REC) {
                    const int flags = __builtin_amdgcn_readfirstlane(recp[1]);
                    float gmask = 0.f, goffw = 0.f, goffh = 0.f;
                    if (flags & 1) {                                           // (uniform)
                        const int base = __builtin_amdgcn_readfirstlane(recp[0]);
                        const int q1 = __builtin_amdgcn_readfirstlane(recp[2]);
                        const float lh = __int_as_float(recp[3]), lw = __int_as_float(recp[4]), mk = __int_as_float(recp[5]);
                        const float hh = 1.f - lh, hw = 1.f - lw;
                        const bool ld_ = !(a.dbg & 8);
                        const float v1 = ((flags & 512) && ld_) ? ET<T>::ld(imb + base) : 0.f;
                        const float v2 = ((flags & 1024) && ld_) ? ET<T>::ld(imb + base + C) : 0.f;
                        const float v3 = ((flags & 2048) && ld_) ? ET<T>::ld(imb + base + rowC) : 0.f;
                        const float v4 = ((flags & 4096) && ld_) ? ET<T>::ld(imb + base + rowC + C) : 0.f;
                        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                        const float val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
                        const float tg = go * mk;
                        auto corner = [&](int ubit, int ibit, float w, int q, int o) {
                            if (flags & ubit) {                                // usable: in the image, weight not zero
                                if (flags & ibit) {                            // inside the window: the column of S (first chunk only)
                                    if (c0 == 0 && lane == (q & 63)) col[q >> 6] += w * mk;
                                } else if (!(a.dbg & 4)) {
                                    atomicAdd(gib + o + lane, w * tg);         // beyond the window: straight to memory, as before
                                }
                            }
                        };
                        corner(2, 32, w1, q1, base);
                        corner(4, 64, w2, q1 + 1, base + C);
                        corner(8, 128, w3, q1 + T_WIN, base + rowC);
                        corner(16, 256, w4, q1 + T_WIN + 1, base + rowC + C);
                        const float ghw = -hw * v1 - lw * v2 + hw * v3 + lw * v4;   // d val / d h
                        const float gww = -hh * v1 + hh * v2 - lh * v3 + lh * v4;   // d val / d w
                        gmask = go * val;
                        goffw = gww * tg;
                        goffh = ghw * tg;
                    }
                    gmask = wave_total63(gmask);
                    goffw = wave_total63(goffw);
                    goffh = wave_total63(goffh);
                    // park the three totals (lane 63) in the lanes that will store them: 18 + 9 contiguous floats per pixel and group
                    const float tm = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gmask), 63));
                    const float tw_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(goffw), 63));
                    const float th_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(goffh), 63));
                    r_msk = lane == k ? tm : r_msk;
                    r_off = lane == 2 * k ? tw_ : (lane == 2 * k + 1 ? th_ : r_off);
                }
                float* gof = a.goff + (size_t)pix * a.G * P * 2 + (size_t)g * P * 2;
                float* gmk = a.gmsk + (size_t)pix * a.G * P + (size_t)g * P;
                if (lane < 2 * P) { if (c0 == 0) gof[lane] = r_off; else gof[lane] += r_off; }      // (same lanes, program order)
                if (lane < P) { if (c0 == 0) gmk[lane] = r_msk; else gmk[lane] += r_msk; }
            }
            // the pixel's column of S (zeros for a pixel beyond the image) and its grad_output row
            if (c0 == 0) {
#pragma unroll
                for (int sl = 0; sl < T_SLOTS; ++sl)
                    if (sl * 64 + lane < T_CPAD) sS[item * T_CPAD + sl * 64 + lane] = col[sl];
            }
            sGO[item * T_GOLD + lane] = go;
        }
        __syncthreads();
        // ---- phase 2: D[cell][c] = sum_pixel S[pixel][cell] * GO[pixel][c]; cell tile mt (16 cells) x 4 channel tiles per wave.
        // A lane ends with cells mt*16 + 4*lgrp + e, channel nt*16 + lrow: an atomic instruction covers 4 cells x 16 channels —
        // four 64-byte pieces, the same four memory-side requests a contiguous 256-byte row makes
        constexpr int NMT = T_CPAD / 16;                      // 21
#pragma unroll 1
        for (int u = 0; u < 2; ++u) {
            const int mt = wave + 16 * u;
            if (mt >= NMT || (a.dbg & 2)) break;              // (uniform)
            f32x4_t d[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) d[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            for (int k4 = 0; k4 < 16; ++k4) {
                const int pxl = k4 * 4 + lgrp;
                const float av = sS[pxl * T_CPAD + mt * 16 + lrow];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    d[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, sGO[pxl * T_GOLD + nt * 16 + lrow], d[nt], 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cell = mt * 16 + 4 * lgrp + e;
                const int ch_h = wh0 + cell / T_WIN, ch_w = ww0 + cell % T_WIN;
                if (cell < T_CELLS && (unsigned)ch_h < (unsigned)a.H && (unsigned)ch_w < (unsigned)a.W) {
                    float* dst = gib + ((size_t)ch_h * a.W + ch_w) * C + lrow;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        if (d[nt][e] != 0.f && !(a.dbg & 1)) atomicAdd(dst + nt * 16, d[nt][e]);
                }
            }
        }
        __syncthreads();                                      // GO (and nothing else) is rewritten by the next chunk
    }
}

// ------------------------------------------------------------------------------------------------------
// Forward, vectorised: a lane owns one 16-byte channel chunk (8 bf16/f16 or 4 f32 channels) of a (pixel, group) item, so
// a 64-lane load instruction moves 1 KiB of gathered rows instead of 128-256 bytes; the offsets and masks of the CTA's
// DCN_PIX consecutive pixels are staged once, coalesced, in LDS as f32 (every lane of an item reads the same 3*P values).
// Same arithmetic as dcnv3_kernel<T, false> (bilinear weights, zero padding, accumulation order over the K*K points).
// ------------------------------------------------------------------------------------------------------
#define DCN_PIX 32
template <typename T>
__global__ __launch_bounds__(256) void dcnv3_fwd_vec_kernel(const DcnArgs a) {
    constexpr int V = ET<T>::V;
    extern __shared__ __attribute__((aligned(16))) float sdcn[];       // [DCN_PIX][G*P*2] offsets, then [DCN_PIX][G*P] masks
    const int P = a.kh * a.kw;
    const int C = a.G * a.Gc;
    const int GP = a.G * P;
    const long long npix = (long long)a.N * a.Ho * a.Wo;
    const int cpi = a.Gc / V;                 // chunks per item
    const int seg = a.seg;                    // lanes per item (power of two >= min(cpi, 64))
    const int ipw = 64 / seg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / seg, cl = lane % seg;
    const T* in = (const T*)a.in;
    float* soff = sdcn;
    float* smsk = sdcn + (size_t)DCN_PIX * GP * 2;
    for (long long pb = (long long)blockIdx.x * DCN_PIX; pb < npix; pb += (long long)gridDim.x * DCN_PIX) {
        const int np = (int)((npix - pb) < DCN_PIX ? (npix - pb) : DCN_PIX);
        __syncthreads();                      // previous tile's readers are done
        for (int i = threadIdx.x; i < np * GP * 2; i += 256) soff[i] = ET<T>::ld((const T*)a.off + (size_t)pb * GP * 2 + i);
        for (int i = threadIdx.x; i < np * GP; i += 256) smsk[i] = ET<T>::ld((const T*)a.msk + (size_t)pb * GP + i);
        __syncthreads();
        const int nitems = np * a.G;
        for (int it0 = wave * ipw; it0 < nitems; it0 += 4 * ipw) {
            const int item = it0 + sub;
            const bool live = item < nitems;
            const int itc = live ? item : 0;
            const int lp = itc / a.G, g = itc - lp * a.G;
            const long long pix = pb + lp;
            const int wo = (int)(pix % a.Wo);
            const long long t2 = pix / a.Wo;
            const int ho = (int)(t2 % a.Ho);
            const int n = (int)(t2 / a.Ho);
            const int p0w = ((a.dw * (a.kw - 1)) >> 1) - a.pw + wo * a.sw;
            const int p0h = ((a.dh * (a.kh - 1)) >> 1) - a.ph + ho * a.sh;
            const float p0w_ = (float)p0w - (float)((a.dw * (a.kw - 1)) >> 1) * a.scale;
            const float p0h_ = (float)p0h - (float)((a.dh * (a.kh - 1)) >> 1) * a.scale;
            const float* offp = soff + ((size_t)lp * a.G + g) * P * 2;
            const float* mskp = smsk + ((size_t)lp * a.G + g) * P;
            const T* imb = in + (size_t)n * a.H * a.W * C + (size_t)g * a.Gc;
            for (int q0 = 0; q0 < cpi; q0 += seg) {
                const int qc = q0 + cl;
                const bool act = live && qc < cpi;
                const int c = qc * V;
                float col[V];
#pragma unroll
                for (int e = 0; e < V; ++e) col[e] = 0.f;
                int k = 0;
                for (int i = 0; i < a.kw; ++i)
                    for (int j = 0; j < a.kh; ++j, ++k) {
                        const float ow = offp[2 * k], oh = offp[2 * k + 1], mk = mskp[k];
                        const float lw_ = p0w_ + ((float)(i * a.dw) + ow) * a.scale;
                        const float lh_ = p0h_ + ((float)(j * a.dh) + oh) * a.scale;
                        if (act && dcn_inside(lh_, lw_, a.H, a.W, a.strict)) {
                            const int hl = (int)floorf(lh_), wl = (int)floorf(lw_);
                            const int hh_ = hl + 1, wh_ = wl + 1;
                            const float lh = lh_ - (float)hl, lw = lw_ - (float)wl;
                            const float hh = 1.f - lh, hw = 1.f - lw;
                            const bool b1 = hl >= 0 && wl >= 0, b2 = hl >= 0 && wh_ <= a.W - 1;
                            const bool b3 = hh_ <= a.H - 1 && wl >= 0, b4 = hh_ <= a.H - 1 && wh_ <= a.W - 1;
                            const uint4 z = make_uint4(0, 0, 0, 0);
                            const uint4 q1 = b1 ? *(const uint4*)(imb + ((size_t)hl * a.W + wl) * C + c) : z;
                            const uint4 q2 = b2 ? *(const uint4*)(imb + ((size_t)hl * a.W + wh_) * C + c) : z;
                            const uint4 q3 = b3 ? *(const uint4*)(imb + ((size_t)hh_ * a.W + wl) * C + c) : z;
                            const uint4 q4 = b4 ? *(const uint4*)(imb + ((size_t)hh_ * a.W + wh_) * C + c) : z;
                            float v1[V], v2[V], v3[V], v4[V];
                            unpack16<T>(q1, v1); unpack16<T>(q2, v2); unpack16<T>(q3, v3); unpack16<T>(q4, v4);
                            const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
#pragma unroll
                            for (int e = 0; e < V; ++e) col[e] += (w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e]) * mk;
                        }
                    }
                if (act) *(uint4*)((T*)a.out + (size_t)pix * C + g * a.Gc + c) = pack16<T>(col);
            }
        }
    }
}

static int fill_args(DcnArgs& a, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w,
                     int dilation_h, int dilation_w, int group, int group_channels, float offset_scale,
                     int N, int H_in, int W_in, int H_out, int W_out) {
    YDL_CHECK(kernel_h > 0 && kernel_w > 0 && stride_h > 0 && stride_w > 0 && group > 0 && group_channels > 0, "bad geometry");
    YDL_CHECK(H_out == (H_in + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) / stride_h + 1 &&
              W_out == (W_in + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) / stride_w + 1, "output size mismatch");
    a.kh = kernel_h; a.kw = kernel_w; a.sh = stride_h; a.sw = stride_w; a.ph = pad_h; a.pw = pad_w;
    a.dh = dilation_h; a.dw = dilation_w; a.G = group; a.Gc = group_channels; a.scale = offset_scale;
    a.N = N; a.H = H_in; a.W = W_in; a.Ho = H_out; a.Wo = W_out;
    int seg = 1;
    while (seg < group_channels && seg < 64) seg <<= 1;
    a.seg = seg;
    a.items = (long long)N * H_out * W_out * group;
    a.strict = g_dcn_border_rule;
    static const int dbg = getenv("YDL_DCN_DBG") ? atoi(getenv("YDL_DCN_DBG")) : 0;
    a.dbg = dbg;
    return 0;
}

static inline int dcn_grid(const DcnArgs& a) {
    long long rounds = (a.items + (64 / a.seg) - 1) / (64 / a.seg);
    long long blocks = (rounds + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

extern "C" int ydl_dcnv3_fwd(int dtype, const void* input, const void* offset, const void* mask, void* output,
                             int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w,
                             int dilation_h, int dilation_w, int group, int group_channels, float offset_scale,
                             int N, int H_in, int W_in, int H_out, int W_out, void* stream) {
    YDL_CHECK(input && offset && mask && output, "null pointer");
    DcnArgs a{};
    if (int e = fill_args(a, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
                          group_channels, offset_scale, N, H_in, W_in, H_out, W_out)) return e;
    a.in = input; a.off = offset; a.msk = mask; a.out = output;
    hipStream_t st = (hipStream_t)stream;
    YDL_CHECK(dtype == YDL_F32 || dtype == YDL_BF16 || dtype == YDL_F16, "bad dtype");
    const int V = dtype == YDL_F32 ? 4 : 8;
    const int P = kernel_h * kernel_w;
    const size_t lds = (size_t)DCN_PIX * group * P * 3 * sizeof(float);
    static const int novec = getenv("YDL_DCN_NOVEC") ? atoi(getenv("YDL_DCN_NOVEC")) : 0;
    if (!novec && group_channels % V == 0 && lds <= 60 * 1024 && aligned16(input) && aligned16(output)) {
        // vectorised path: 16-byte chunks, offsets/masks staged in LDS
        int seg = 1;
        while (seg < group_channels / V && seg < 64) seg <<= 1;
        a.seg = seg;
        const long long npix = (long long)N * H_out * W_out;
        long long blocks = (npix + DCN_PIX - 1) / DCN_PIX;
        if (blocks > 256 * 8) blocks = 256 * 8;
        if (dtype == YDL_F32) dcnv3_fwd_vec_kernel<float><<<(int)blocks, 256, lds, st>>>(a);
        else if (dtype == YDL_BF16) dcnv3_fwd_vec_kernel<bf16_t><<<(int)blocks, 256, lds, st>>>(a);
        else dcnv3_fwd_vec_kernel<_Float16><<<(int)blocks, 256, lds, st>>>(a);
        YDL_LAUNCH_CHECK();
        return 0;
    }
    if (dtype == YDL_F32) dcnv3_kernel<float, false><<<dcn_grid(a), 256, 0, st>>>(a);
    else if (dtype == YDL_BF16) dcnv3_kernel<bf16_t, false><<<dcn_grid(a), 256, 0, st>>>(a);
    else dcnv3_kernel<_Float16, false><<<dcn_grid(a), 256, 0, st>>>(a);
    YDL_LAUNCH_CHECK();
    return 0;
}

extern "C" int ydl_dcnv3_bwd(int dtype, const void* input, const void* offset, const void* mask, const void* grad_output,
                             float* grad_input, float* grad_offset, float* grad_mask,
                             int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w,
                             int dilation_h, int dilation_w, int group, int group_channels, float offset_scale,
                             int N, int H_in, int W_in, int H_out, int W_out, void* stream) {
    YDL_CHECK(input && offset && mask && grad_output && grad_input && grad_offset && grad_mask, "null pointer");
    DcnArgs a{};
    if (int e = fill_args(a, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
                          group_channels, offset_scale, N, H_in, W_in, H_out, W_out)) return e;
    a.in = input; a.off = offset; a.msk = mask; a.gout = grad_output;
    a.gin = grad_input; a.goff = grad_offset; a.gmsk = grad_mask;
    hipStream_t st = (hipStream_t)stream;
    YDL_CHECK(dtype == YDL_F32 || dtype == YDL_BF16 || dtype == YDL_F16, "bad dtype");
    static const int nowin = getenv("YDL_DCN_NOWIN") ? atoi(getenv("YDL_DCN_NOWIN")) : 0;
    static const int notile = getenv("YDL_DCN_NOTILE") ? atoi(getenv("YDL_DCN_NOTILE")) : 0;
    if (!nowin && !notile && g_dcn_win && g_dcn_tile && kernel_h == 3 && kernel_w == 3 && group_channels % 64 == 0 && stride_h == 1 && stride_w == 1 &&
        dilation_h == 1 && dilation_w == 1 && offset_scale == 1.0f && H_out >= 8 && W_out >= 8 &&
        (g_dcn_tile == 2 || (long long)N * group * ((W_out + 7) / 8) * ((H_out + 7) / 8) >= 2ll * ydl_device_cus())) {
        // (a CTA per tile and group, one per CU: maps with fewer than two rounds of tiles — 20 x 20 at batch 16 — keep the
        //  wave-per-pixel kernels: 145 / 203 us against 244 / 295 us at sigma 0 / 2)
        // 8 x 8 pixel tiles: the scatter into grad_input as S x grad_output on the f32 MFMA (dcnv3_bwd_tile_kernel)
        const int tiles_w = (W_out + 7) / 8, tiles_hw = tiles_w * ((H_out + 7) / 8);
        const long long blocks = (long long)N * group * tiles_hw;
        YDL_CHECK(blocks < (1ll << 31), "too many tiles");
        const size_t lds = (size_t)(64 * T_CPAD + 64 * T_GOLD + 64 * 9 * T_REC) * sizeof(float);
        if (dtype == YDL_F32) { YDL_SET_MAX_LDS((dcnv3_bwd_tile_kernel<float>), lds); dcnv3_bwd_tile_kernel<float><<<(int)blocks, 1024, lds, st>>>(a, tiles_w, tiles_hw); }
        else if (dtype == YDL_BF16) { YDL_SET_MAX_LDS((dcnv3_bwd_tile_kernel<bf16_t>), lds); dcnv3_bwd_tile_kernel<bf16_t><<<(int)blocks, 1024, lds, st>>>(a, tiles_w, tiles_hw); }
        else { YDL_SET_MAX_LDS((dcnv3_bwd_tile_kernel<_Float16>), lds); dcnv3_bwd_tile_kernel<_Float16><<<(int)blocks, 1024, lds, st>>>(a, tiles_w, tiles_hw); }
        YDL_LAUNCH_CHECK();
        return 0;
    }
    if (!nowin && g_dcn_win && kernel_h == 3 && kernel_w == 3 && a.seg == 64) {
        // a whole wave per item: corner gradients merged in a register window before the atomics (dcnv3_bwd_win_kernel)
        long long blocks = (a.items + 3) / 4;
        if (blocks > 256 * 8) blocks = 256 * 8;
        if (dtype == YDL_F32) dcnv3_bwd_win_kernel<float><<<(int)blocks, 256, 0, st>>>(a);
        else if (dtype == YDL_BF16) dcnv3_bwd_win_kernel<bf16_t><<<(int)blocks, 256, 0, st>>>(a);
        else dcnv3_bwd_win_kernel<_Float16><<<(int)blocks, 256, 0, st>>>(a);
        YDL_LAUNCH_CHECK();
        return 0;
    }
    if (dtype == YDL_F32) dcnv3_kernel<float, true><<<dcn_grid(a), 256, 0, st>>>(a);
    else if (dtype == YDL_BF16) dcnv3_kernel<bf16_t, true><<<dcn_grid(a), 256, 0, st>>>(a);
    else dcnv3_kernel<_Float16, true><<<dcn_grid(a), 256, 0, st>>>(a);
    YDL_LAUNCH_CHECK();
    return 0;
}
