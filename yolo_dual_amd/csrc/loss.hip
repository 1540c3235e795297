// Channel softmax (the yaml models end in nn.Softmax(1)), the fused CE + 0.5*(Dice|Jaccard) loss with its
// backward, and the argmax + confusion-matrix evaluator.  One thread per pixel; per-(n,c) sums are reduced with
// wavefront shuffles -> LDS -> per-block partials -> a tiny double-precision merge (deterministic, no atomics).
#include "common.h"

#define MAXC 32

// ------------------------------------------------------------------------------------------------------
// softmax over C (<= 32) channels of an NHWC tensor -> f32 tensor with arbitrary strides.
// (rh, rw) = nearest-replication factors: the output/gradient tensor is (N, C, H*rh, W*rw) while x is stored at
// (H, W) — the lazily up-sampled point-wise tail (softmax commutes with nearest up-sampling).
//   forward : one thread per OUTPUT pixel (recomputes the 12 exps; fully coalesced stores)
//   backward: one thread per STORED pixel, sums dp over its rh x rw replicas, then dx = p*(dps - sum_c p*dps)
// ------------------------------------------------------------------------------------------------------
template <typename T, int MC>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const T* __restrict__ x, int ldx, float* __restrict__ p,
                                                          long long sn, long long sc, long long sh, long long sw,
                                                          int N, int H, int W, int C, int rh, int rw) {
    const int LH = H * rh, LW = W * rw;
    const long long total = (long long)N * LH * LW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int w = (int)(i % LW);
        long long t2 = i / LW;
        int h = (int)(t2 % LH);
        int n = (int)(t2 / LH);
        const T* xp = x + ((size_t)(n * H + h / rh) * W + w / rw) * ldx;
        float v[MC];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) { v[c] = ET<T>::ld(xp + c); mx = fmaxf(mx, v[c]); }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) { v[c] = expf(v[c] - mx); s += v[c]; }
        float inv = 1.f / s;
        float* pp = p + n * sn + h * sh + w * sw;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) pp[c * sc] = v[c] * inv;
    }
}

// replicated output, rw == 4, unit W stride: one thread per STORED pixel computes the softmax once and writes its rh x 4 replicas as
// 16-byte stores (a wave-instruction covers 1 KiB of one output row); the generic kernel recomputed the 12 exponentials for each
// of the 16 replicas and was bound by them (97 us for the 315 MB export of BASELINE config 2, 58 us here)
template <typename T, int MC>
__global__ __launch_bounds__(256) void softmax_fwd_rep4_kernel(const T* __restrict__ x, int ldx, float* __restrict__ p,
                                                               long long sn, long long sc, long long sh,
                                                               int N, int H, int W, int C, int rh) {
    const long long total = (long long)N * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const long long t2 = i / W;
        const int h = (int)(t2 % H);
        const int n = (int)(t2 / H);
        const T* xp = x + (size_t)i * ldx;
        float v[MC];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) { v[c] = ET<T>::ld(xp + c); mx = fmaxf(mx, v[c]); }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) { v[c] = expf(v[c] - mx); s += v[c]; }
        const float inv = 1.f / s;
        float* pp = p + n * sn + (long long)(h * rh) * sh + (long long)w * 4;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) {
                const float q = v[c] * inv;
                const float4 q4 = make_float4(q, q, q, q);
                for (int a = 0; a < rh; ++a) *(float4*)(pp + c * sc + a * sh) = q4;
            }
    }
}

template <typename T, int MC>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                          long long sn, long long sc, long long sh, long long sw,
                                                          T* __restrict__ dx, int lddx, int N, int H, int W, int C, int Cp,
                                                          int rh, int rw) {
    const long long total = (long long)N * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int w = (int)(i % W);
        long long t2 = i / W;
        int h = (int)(t2 % H);
        int n = (int)(t2 / H);
        const long long off = n * sn + (long long)(h * rh) * sh + (long long)(w * rw) * sw;
        float pv[MC], gv[MC];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) {
                pv[c] = p[off + c * sc];
                float g = 0.f;
                for (int a = 0; a < rh; ++a)
                    for (int b = 0; b < rw; ++b) g += dp[off + c * sc + a * sh + b * sw];
                gv[c] = g;
                dot += pv[c] * g;
            }
        T* d = dx + (size_t)i * lddx;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < Cp) ET<T>::st(d + c, c < C ? pv[c] * (gv[c] - dot) : 0.f);
    }
}

static inline int sgrid(long long total) {
    long long b = (total + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int ydl_softmax_fwd(int dtype, const void* x, int ldx, float* p, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                               int N, int H, int W, int C, int rep_h, int rep_w, void* stream) {
    YDL_CHECK(x && p && C >= 1 && C <= MAXC && ldx >= C && rep_h >= 1 && rep_w >= 1, "bad arguments (C <= 32)");
    hipStream_t st = (hipStream_t)stream;
    if (rep_w == 4 && sw == 1 && C <= 16 && ((uintptr_t)p & 15) == 0 && sn % 4 == 0 && sc % 4 == 0 && sh % 4 == 0) {
        const int g2 = sgrid((long long)N * H * W);
        if (dtype == YDL_F32) softmax_fwd_rep4_kernel<float, 16><<<g2, 256, 0, st>>>((const float*)x, ldx, p, sn, sc, sh, N, H, W, C, rep_h);
        else softmax_fwd_rep4_kernel<bf16_t, 16><<<g2, 256, 0, st>>>((const bf16_t*)x, ldx, p, sn, sc, sh, N, H, W, C, rep_h);
        YDL_LAUNCH_CHECK();
        return 0;
    }
    int grid = sgrid((long long)N * H * W * rep_h * rep_w);
    if (dtype == YDL_F32) {
        if (C <= 16) softmax_fwd_kernel<float, 16><<<grid, 256, 0, st>>>((const float*)x, ldx, p, sn, sc, sh, sw, N, H, W, C, rep_h, rep_w);
        else softmax_fwd_kernel<float, 32><<<grid, 256, 0, st>>>((const float*)x, ldx, p, sn, sc, sh, sw, N, H, W, C, rep_h, rep_w);
    } else {
        if (C <= 16) softmax_fwd_kernel<bf16_t, 16><<<grid, 256, 0, st>>>((const bf16_t*)x, ldx, p, sn, sc, sh, sw, N, H, W, C, rep_h, rep_w);
        else softmax_fwd_kernel<bf16_t, 32><<<grid, 256, 0, st>>>((const bf16_t*)x, ldx, p, sn, sc, sh, sw, N, H, W, C, rep_h, rep_w);
    }
    YDL_LAUNCH_CHECK();
    return 0;
}
extern "C" int ydl_softmax_bwd(int dtype, const float* p, const float* dp, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                               void* dx, int lddx, int N, int H, int W, int C, int rep_h, int rep_w, void* stream) {
    const int Cp = round_up(C, 8) <= lddx ? round_up(C, 8) : C;
    YDL_CHECK(p && dp && dx && C >= 1 && C <= MAXC && lddx >= C && rep_h >= 1 && rep_w >= 1, "bad arguments (C <= 32)");
    hipStream_t st = (hipStream_t)stream;
    int grid = sgrid((long long)N * H * W);
    if (dtype == YDL_F32) {
        if (Cp <= 16) softmax_bwd_kernel<float, 16><<<grid, 256, 0, st>>>(p, dp, sn, sc, sh, sw, (float*)dx, lddx, N, H, W, C, Cp, rep_h, rep_w);
        else softmax_bwd_kernel<float, 32><<<grid, 256, 0, st>>>(p, dp, sn, sc, sh, sw, (float*)dx, lddx, N, H, W, C, Cp, rep_h, rep_w);
    } else {
        if (Cp <= 16) softmax_bwd_kernel<bf16_t, 16><<<grid, 256, 0, st>>>(p, dp, sn, sc, sh, sw, (bf16_t*)dx, lddx, N, H, W, C, Cp, rep_h, rep_w);
        else softmax_bwd_kernel<bf16_t, 32><<<grid, 256, 0, st>>>(p, dp, sn, sc, sh, sw, (bf16_t*)dx, lddx, N, H, W, C, Cp, rep_h, rep_w);
    }
    YDL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// segmentation loss.  total = CE(weight, label_smoothing)(pred, t) + 0.5 * (1 - mean_{n,c} R[n,c])
//   p = softmax(pred);  I = sum cw_c p_c [t=c];  P = sum cw_c p_c;  T = sum [t=c]
//   dice: R = (2I+eps)/(P+T+eps)     jaccard: R = (I+eps)/(P+T-I+eps)
//   CE  = (1-ls) * sum_i w_t (-logp_t) / sum_i w_t  +  ls/C * sum_i sum_c w_c (-logp_c) / sum_i w_t
// ws layout (floats), NC = N*C, nblk = LOSS_BLOCKS_PER_IMAGE:
//   [0, NC)        I      [NC, 2NC) P     [2NC, 3NC) T        (merged sums)
//   [3NC, 4NC)     aI = d ov/d I   [4NC, 5NC) aP = d ov/d P   (backward coefficients)
//   [5NC, 5NC+4)   ce_num, ce_den, smooth_num, W = sum_c w_c
//   [5NC+4, ...)   per-block partials [N][nblk][3C+3], then per-image {sumR, ce_num, ce_den, sm_num} [N][4]
// ------------------------------------------------------------------------------------------------------
#define LOSS_BLOCKS_PER_IMAGE 128

extern "C" int64_t ydl_seg_loss_ws_floats(int N, int C) {
    return (int64_t)5 * N * C + 4 + (int64_t)N * LOSS_BLOCKS_PER_IMAGE * (3 * C + 3) + (int64_t)4 * N;
}

__device__ __forceinline__ int target_at(const int64_t* __restrict__ target, int n, int h, int w, int H, int W, int Ht, int Wt,
                                         float sth, float stw) {
    int th = h, tw = w;
    if (Ht != H || Wt != W) {   // F.interpolate(mode='nearest') of the label map
        th = (int)floorf(__fmul_rn((float)h, sth)); if (th > Ht - 1) th = Ht - 1;
        tw = (int)floorf(__fmul_rn((float)w, stw)); if (tw > Wt - 1) tw = Wt - 1;
    }
    return (int)target[((size_t)n * Ht + th) * Wt + tw];
}

template <int MC>
__global__ __launch_bounds__(256) void seg_loss_fwd_kernel(const float* __restrict__ pred, long long sn, long long sc, long long sh,
                                                           long long sw, const int64_t* __restrict__ target, int Ht, int Wt,
                                                           const float* __restrict__ cw, int C, int H, int W,
                                                           float sth, float stw, float* __restrict__ part) {
    const int n = blockIdx.y;
    const long long HW = (long long)H * W;
    float accI[MC], accP[MC], accT[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) { accI[c] = 0.f; accP[c] = 0.f; accT[c] = 0.f; }
    float ce_num = 0.f, ce_den = 0.f, sm_num = 0.f;
    float wv[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) wv[c] = (c < C) ? (cw ? cw[c] : 1.f) : 0.f;

    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x) {
        int w = (int)(i % W), h = (int)(i / W);
        const float* pp = pred + n * sn + h * sh + w * sw;
        int t = target_at(target, n, h, w, H, W, Ht, Wt, sth, stw);
        float v[MC];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) { v[c] = pp[c * sc]; mx = fmaxf(mx, v[c]); }
        float s = 0.f;
        float ex[MC];
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) { ex[c] = expf(v[c] - mx); s += ex[c]; }
        float lse = mx + logf(s);
        float inv = 1.f / s;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) {
                float pc = ex[c] * inv;
                float wp = wv[c] * pc;
                bool hit = (c == t);
                accP[c] += wp;
                accI[c] += hit ? wp : 0.f;
                accT[c] += hit ? 1.f : 0.f;
                float nlp = lse - v[c];          // -log p_c
                sm_num += wv[c] * nlp;
                if (hit) { ce_num += wv[c] * nlp; ce_den += wv[c]; }
            }
    }
    // block reduction: wave shuffles then LDS across the 4 waves
    __shared__ float red[4][3 * MC + 3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < MC; ++c)
        if (c < C) {
            float a = wave_total63(accI[c]), b = wave_total63(accP[c]), d = wave_total63(accT[c]);      // (DPP: totals in lane 63)
            if (lane == 63) { red[wave][c] = a; red[wave][MC + c] = b; red[wave][2 * MC + c] = d; }
        }
    {
        float a = wave_total63(ce_num), b = wave_total63(ce_den), d = wave_total63(sm_num);
        if (lane == 63) { red[wave][3 * MC] = a; red[wave][3 * MC + 1] = b; red[wave][3 * MC + 2] = d; }
    }
    __syncthreads();
    float* dst = part + ((size_t)n * gridDim.x + blockIdx.x) * (3 * C + 3);
    if (threadIdx.x < 3 * C + 3) {
        int k = threadIdx.x;
        int src = k < 3 * C ? (k / C) * MC + (k % C) : 3 * MC + (k - 3 * C);
        dst[k] = red[0][src] + red[1][src] + red[2][src] + red[3][src];
    }
}

// merge level 1: one CTA per image; 8 slices x 32 channels walk the per-CTA partials (16 steps each), LDS-reduce,
// write I/P/T, the overlap ratio R and its derivatives for this image's (n, c) pairs, and per-image CE partial sums.
// ws tail used as scratch: [5NC+4+N*nblk*(3C+3) ...) holds per-image {sumR, ce_num, ce_den, sm_num}.
__global__ __launch_bounds__(256) void seg_loss_merge1_kernel(float* __restrict__ ws, int N, int C, int nblk, int kind,
                                                              float eps, float* __restrict__ img_part) {
    const int n = blockIdx.x;
    const int NC = N * C;
    const float* part = ws + 5 * NC + 4 + (size_t)n * nblk * (3 * C + 3);
    __shared__ double sh[3][8][33];
    __shared__ double sR[32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    double I = 0, P = 0, T = 0;
    if (cl < C)
        for (int b = sl; b < nblk; b += 8) {
            const float* pb = part + (size_t)b * (3 * C + 3);
            I += pb[cl]; P += pb[C + cl]; T += pb[2 * C + cl];
        }
    sh[0][sl][cl] = I; sh[1][sl][cl] = P; sh[2][sl][cl] = T;
    __syncthreads();
    if (sl == 0) {
        double R = 0.0;
        if (cl < C) {
            I = P = T = 0;
            for (int i = 0; i < 8; ++i) { I += sh[0][i][cl]; P += sh[1][i][cl]; T += sh[2][i][cl]; }
            const int k = n * C + cl;
            ws[k] = (float)I; ws[NC + k] = (float)P; ws[2 * NC + k] = (float)T;
            double aI, aP;
            if (kind == YDL_LOSS_DICE) {
                double den = P + T + eps;
                R = (2.0 * I + eps) / den;
                aI = 2.0 / den;
                aP = -(2.0 * I + eps) / (den * den);
            } else {
                double num = I + eps, den = P + T - I + eps;
                R = num / den;
                aI = 1.0 / den + num / (den * den);
                aP = -num / (den * den);
            }
            ws[3 * NC + k] = (float)(-aI / NC);     // d ov / d I   (ov = 1 - mean R)
            ws[4 * NC + k] = (float)(-aP / NC);
        }
        sR[cl] = R;
    }
    __syncthreads();
    // CE partial sums of this image: threads 0..2 of slice 1 would walk nblk entries; spread over the whole CTA
    double ce[3] = {0, 0, 0};
    for (int b = threadIdx.x; b < nblk; b += 256) {
        const float* pb = part + (size_t)b * (3 * C + 3) + 3 * C;
        ce[0] += pb[0]; ce[1] += pb[1]; ce[2] += pb[2];
    }
    __shared__ double sce[3][4];
    for (int j = 0; j < 3; ++j) {
        double v = wave_sum_d(ce[j]);
        if ((threadIdx.x & 63) == 0) sce[j][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0;
        for (int c = 0; c < C; ++c) r += sR[c];
        img_part[n * 4 + 0] = (float)r;
        for (int j = 0; j < 3; ++j) img_part[n * 4 + 1 + j] = (float)(sce[j][0] + sce[j][1] + sce[j][2] + sce[j][3]);
    }
}

__global__ void seg_loss_merge2_kernel(float* __restrict__ ws, int N, int C, float ls, const float* __restrict__ cw,
                                       const float* __restrict__ img_part, float* __restrict__ losses) {
    if (threadIdx.x != 0) return;
    const int NC = N * C;
    double r = 0, a = 0, b = 0, d = 0;
    for (int n = 0; n < N; ++n) { r += img_part[n * 4]; a += img_part[n * 4 + 1]; b += img_part[n * 4 + 2]; d += img_part[n * 4 + 3]; }
    double ov = 1.0 - r / NC;
    double Wsum = 0;
    for (int c = 0; c < C; ++c) Wsum += cw ? cw[c] : 1.0;
    double ce = (1.0 - ls) * a / b + (ls > 0.f ? (double)ls / C * d / b : 0.0);
    ws[5 * NC + 0] = (float)a; ws[5 * NC + 1] = (float)b; ws[5 * NC + 2] = (float)d; ws[5 * NC + 3] = (float)Wsum;
    losses[0] = (float)(ce + 0.5 * ov);
    losses[1] = (float)ce;
    losses[2] = (float)ov;
}

template <int MC>
__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const float* __restrict__ pred, long long sn, long long sc, long long sh,
                                                           long long sw, const int64_t* __restrict__ target, int Ht, int Wt,
                                                           const float* __restrict__ cw, int C, int H, int W, float sth, float stw,
                                                           float ls, const float* __restrict__ ws, int N,
                                                           const float* __restrict__ dloss, float* __restrict__ dpred) {
    const int n = blockIdx.y;
    const int NC = N * C;
    const long long HW = (long long)H * W;
    const float g0 = dloss ? dloss[0] : 1.f;
    const float ce_den = ws[5 * NC + 1], Wsum = ws[5 * NC + 3];
    float wv[MC], aI[MC], aP[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        wv[c] = (c < C) ? (cw ? cw[c] : 1.f) : 0.f;
        aI[c] = (c < C) ? ws[3 * NC + n * C + c] : 0.f;
        aP[c] = (c < C) ? ws[4 * NC + n * C + c] : 0.f;
    }
    const float k_nll = (1.f - ls) / ce_den;
    const float k_sm = ls > 0.f ? ls / ((float)C * ce_den) : 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x) {
        int w = (int)(i % W), h = (int)(i / W);
        const long long off = n * sn + h * sh + w * sw;
        int t = target_at(target, n, h, w, H, W, Ht, Wt, sth, stw);
        float v[MC];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) { v[c] = pred[off + c * sc]; mx = fmaxf(mx, v[c]); }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) { v[c] = expf(v[c] - mx); s += v[c]; }
        float inv = 1.f / s;
        float gdot = 0.f;
        float g[MC];
        float wt = 0.f;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) {
                v[c] *= inv;                                   // p_c
                g[c] = wv[c] * (aP[c] + (c == t ? aI[c] : 0.f));   // d ov / d p_c
                gdot += g[c] * v[c];
                if (c == t) wt = wv[c];
            }
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) {
                float hit = (c == t) ? 1.f : 0.f;
                float d_ce = k_nll * wt * (v[c] - hit) + k_sm * (v[c] * Wsum - wv[c]);
                float d_ov = v[c] * (g[c] - gdot);
                dpred[off + c * sc] = g0 * (d_ce + 0.5f * d_ov);
            }
    }
}

extern "C" int ydl_seg_loss_fwd(const float* pred, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                                const int64_t* target, int Ht, int Wt, const float* class_weights,
                                int kind, float label_smoothing, float eps, int N, int C, int H, int W,
                                float* ws, float* losses, void* stream) {
    YDL_CHECK(pred && target && ws && losses, "null pointer");
    YDL_CHECK(C >= 1 && C <= MAXC && N >= 1 && H >= 1 && W >= 1 && Ht >= 1 && Wt >= 1, "bad sizes (C <= 32)");
    YDL_CHECK(kind == YDL_LOSS_DICE || kind == YDL_LOSS_JACCARD, "bad loss kind");
    hipStream_t st = (hipStream_t)stream;
    float sth = (float)Ht / (float)H, stw = (float)Wt / (float)W;
    dim3 grid(LOSS_BLOCKS_PER_IMAGE, N);
    if (C <= 16)
        seg_loss_fwd_kernel<16><<<grid, 256, 0, st>>>(pred, sn, sc, sh, sw, target, Ht, Wt, class_weights, C, H, W, sth, stw,
                                                      ws + (size_t)5 * N * C + 4);
    else
        seg_loss_fwd_kernel<32><<<grid, 256, 0, st>>>(pred, sn, sc, sh, sw, target, Ht, Wt, class_weights, C, H, W, sth, stw,
                                                      ws + (size_t)5 * N * C + 4);
    float* img_part = ws + (size_t)5 * N * C + 4 + (size_t)N * LOSS_BLOCKS_PER_IMAGE * (3 * C + 3);
    seg_loss_merge1_kernel<<<N, 256, 0, st>>>(ws, N, C, LOSS_BLOCKS_PER_IMAGE, kind, eps, img_part);
    seg_loss_merge2_kernel<<<1, 64, 0, st>>>(ws, N, C, label_smoothing, class_weights, img_part, losses);
    YDL_LAUNCH_CHECK();
    return 0;
}

extern "C" int ydl_seg_loss_bwd(const float* pred, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                                const int64_t* target, int Ht, int Wt, const float* class_weights,
                                int kind, float label_smoothing, float eps, int N, int C, int H, int W,
                                const float* ws, const float* dloss, float* dpred, void* stream) {
    (void)kind; (void)eps;
    YDL_CHECK(pred && target && ws && dpred, "null pointer");
    YDL_CHECK(C >= 1 && C <= MAXC, "C <= 32");
    hipStream_t st = (hipStream_t)stream;
    float sth = (float)Ht / (float)H, stw = (float)Wt / (float)W;
    long long HW = (long long)H * W;
    int bx = (int)((HW + 255) / 256);
    if (bx > 1024) bx = 1024;
    dim3 grid(bx, N);
    if (C <= 16)
        seg_loss_bwd_kernel<16><<<grid, 256, 0, st>>>(pred, sn, sc, sh, sw, target, Ht, Wt, class_weights, C, H, W, sth, stw,
                                                      label_smoothing, ws, N, dloss, dpred);
    else
        seg_loss_bwd_kernel<32><<<grid, 256, 0, st>>>(pred, sn, sc, sh, sw, target, Ht, Wt, class_weights, C, H, W, sth, stw,
                                                      label_smoothing, ws, N, dloss, dpred);
    YDL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// Replicated-prediction variant.  The yaml models end in  Upsample(nearest, r) -> Conv 1x1 -> Softmax, which this
// library evaluates lazily at the stored resolution: the (N, C, H*rh, W*rw) prediction is an exact rh x rw
// replication of `plow` (N, C, H, W).  Every per-pixel term of the loss then depends on the label only, so one thread
// per STORED pixel reads its rh*rw labels, counts them per class and adds count-weighted terms: the same sums as the
// full-resolution kernels above (re-associated), at 1/(rh*rw) of the prediction traffic.  The backward returns
// dlow[n,c,h,w] = sum over the replicas of d loss / d pred, which is what the lazy up-sampling backward needs.
// ------------------------------------------------------------------------------------------------------
template <int MC>
__device__ __forceinline__ void count_labels(const int64_t* __restrict__ target, int n, int h, int w, int Ht, int Wt, int rh, int rw,
                                             float (&cnt)[MC]) {
#pragma unroll
    for (int c = 0; c < MC; ++c) cnt[c] = 0.f;
    const int64_t* tp = target + ((size_t)n * Ht + (size_t)h * rh) * Wt + (size_t)w * rw;
    if (rw == 4 && (Wt & 1) == 0) {
        for (int a = 0; a < rh; ++a) {
            const longlong2* q = (const longlong2*)(tp + (size_t)a * Wt);
            longlong2 u0 = q[0], u1 = q[1];
            int t0 = (int)u0.x, t1 = (int)u0.y, t2 = (int)u1.x, t3 = (int)u1.y;
#pragma unroll
            for (int c = 0; c < MC; ++c)
                cnt[c] += (float)((t0 == c) + (t1 == c) + (t2 == c) + (t3 == c));
        }
    } else {
        for (int a = 0; a < rh; ++a)
            for (int b = 0; b < rw; ++b) {
                int t = (int)tp[(size_t)a * Wt + b];
#pragma unroll
                for (int c = 0; c < MC; ++c) cnt[c] += (t == c) ? 1.f : 0.f;
            }
    }
}

template <int MC>
__global__ __launch_bounds__(256) void seg_loss_rep_fwd_kernel(const float* __restrict__ plow, long long sn, long long sc, long long sh,
                                                               long long sw, const int64_t* __restrict__ target,
                                                               const float* __restrict__ cw, int C, int H, int W, int rh, int rw,
                                                               float* __restrict__ part) {
    const int n = blockIdx.y;
    const long long HW = (long long)H * W;
    const int Ht = H * rh, Wt = W * rw;
    const float rr = (float)(rh * rw);
    float accI[MC], accP[MC], accT[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) { accI[c] = 0.f; accP[c] = 0.f; accT[c] = 0.f; }
    float ce_num = 0.f, ce_den = 0.f, sm_num = 0.f;
    float wv[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) wv[c] = (c < C) ? (cw ? cw[c] : 1.f) : 0.f;

    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x) {
        int w = (int)(i % W), h = (int)(i / W);
        const float* pp = plow + n * sn + h * sh + w * sw;
        float cnt[MC];
        count_labels<MC>(target, n, h, w, Ht, Wt, rh, rw, cnt);
        float v[MC];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) { v[c] = pp[c * sc]; mx = fmaxf(mx, v[c]); }
        float s = 0.f;
        float ex[MC];
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) { ex[c] = expf(v[c] - mx); s += ex[c]; }
        float lse = mx + logf(s);
        float inv = 1.f / s;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) {
                float pc = ex[c] * inv;
                float wp = wv[c] * pc;
                accP[c] += rr * wp;
                accI[c] += cnt[c] * wp;
                accT[c] += cnt[c];
                float wn = wv[c] * (lse - v[c]);     // w_c * (-log p_c)
                sm_num += rr * wn;
                ce_num += cnt[c] * wn;
                ce_den += cnt[c] * wv[c];
            }
    }
    __shared__ float red[4][3 * MC + 3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < MC; ++c)
        if (c < C) {
            float a = wave_total63(accI[c]), b = wave_total63(accP[c]), d = wave_total63(accT[c]);      // (DPP: totals in lane 63)
            if (lane == 63) { red[wave][c] = a; red[wave][MC + c] = b; red[wave][2 * MC + c] = d; }
        }
    {
        float a = wave_total63(ce_num), b = wave_total63(ce_den), d = wave_total63(sm_num);
        if (lane == 63) { red[wave][3 * MC] = a; red[wave][3 * MC + 1] = b; red[wave][3 * MC + 2] = d; }
    }
    __syncthreads();
    float* dst = part + ((size_t)n * gridDim.x + blockIdx.x) * (3 * C + 3);
    if (threadIdx.x < 3 * C + 3) {
        int k = threadIdx.x;
        int src = k < 3 * C ? (k / C) * MC + (k % C) : 3 * MC + (k - 3 * C);
        dst[k] = red[0][src] + red[1][src] + red[2][src] + red[3][src];
    }
}

// sum over the rh*rw replicas of the full-resolution gradient (labels enter through their per-class counts):
//   S = sum_t cnt_t w_t,  A = sum_t cnt_t w_t aI_t p_t,  G0 = sum_c w_c aP_c p_c,  rr = rh*rw
//   dlow_c = g0 * [ k_nll (p_c S - cnt_c w_c) + k_sm rr (p_c Wsum - w_c) + 0.5 p_c (rr w_c aP_c + cnt_c w_c aI_c - rr G0 - A) ]
template <int MC>
__global__ __launch_bounds__(256) void seg_loss_rep_bwd_kernel(const float* __restrict__ plow, long long sn, long long sc, long long sh,
                                                               long long sw, const int64_t* __restrict__ target,
                                                               const float* __restrict__ cw, int C, int H, int W, int rh, int rw,
                                                               float ls, const float* __restrict__ ws, int N,
                                                               const float* __restrict__ dloss, float* __restrict__ dlow) {
    const int n = blockIdx.y;
    const int NC = N * C;
    const long long HW = (long long)H * W;
    const int Ht = H * rh, Wt = W * rw;
    const float rr = (float)(rh * rw);
    const float g0 = dloss ? dloss[0] : 1.f;
    const float ce_den = ws[5 * NC + 1], Wsum = ws[5 * NC + 3];
    float wv[MC], aI[MC], aP[MC];
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        wv[c] = (c < C) ? (cw ? cw[c] : 1.f) : 0.f;
        aI[c] = (c < C) ? ws[3 * NC + n * C + c] : 0.f;
        aP[c] = (c < C) ? ws[4 * NC + n * C + c] : 0.f;
    }
    const float k_nll = (1.f - ls) / ce_den;
    const float k_sm = ls > 0.f ? ls / ((float)C * ce_den) : 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long long)gridDim.x * blockDim.x) {
        int w = (int)(i % W), h = (int)(i / W);
        const long long off = n * sn + h * sh + w * sw;
        float cnt[MC];
        count_labels<MC>(target, n, h, w, Ht, Wt, rh, rw, cnt);
        float v[MC];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) { v[c] = plow[off + c * sc]; mx = fmaxf(mx, v[c]); }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) { v[c] = expf(v[c] - mx); s += v[c]; }
        float inv = 1.f / s;
        float S = 0.f, A = 0.f, G0 = 0.f;
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) {
                v[c] *= inv;
                float cwc = cnt[c] * wv[c];
                S += cwc;
                A += cwc * aI[c] * v[c];
                G0 += wv[c] * aP[c] * v[c];
            }
#pragma unroll
        for (int c = 0; c < MC; ++c)
            if (c < C) {
                float cwc = cnt[c] * wv[c];
                float d_ce = k_nll * (v[c] * S - cwc) + k_sm * rr * (v[c] * Wsum - wv[c]);
                float d_ov = v[c] * (rr * wv[c] * aP[c] + cwc * aI[c] - rr * G0 - A);
                dlow[off + c * sc] = g0 * (d_ce + 0.5f * d_ov);
            }
    }
}

extern "C" int ydl_seg_loss_rep_fwd(const float* plow, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                                    const int64_t* target, const float* class_weights, int kind, float label_smoothing, float eps,
                                    int N, int C, int H, int W, int rep_h, int rep_w, float* ws, float* losses, void* stream) {
    YDL_CHECK(plow && target && ws && losses, "null pointer");
    YDL_CHECK(C >= 1 && C <= MAXC && N >= 1 && H >= 1 && W >= 1 && rep_h >= 1 && rep_w >= 1, "bad sizes (C <= 32)");
    YDL_CHECK(kind == YDL_LOSS_DICE || kind == YDL_LOSS_JACCARD, "bad loss kind");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(LOSS_BLOCKS_PER_IMAGE, N);
    float* part = ws + (size_t)5 * N * C + 4;
    if (C <= 16)
        seg_loss_rep_fwd_kernel<16><<<grid, 256, 0, st>>>(plow, sn, sc, sh, sw, target, class_weights, C, H, W, rep_h, rep_w, part);
    else
        seg_loss_rep_fwd_kernel<32><<<grid, 256, 0, st>>>(plow, sn, sc, sh, sw, target, class_weights, C, H, W, rep_h, rep_w, part);
    float* img_part = part + (size_t)N * LOSS_BLOCKS_PER_IMAGE * (3 * C + 3);
    seg_loss_merge1_kernel<<<N, 256, 0, st>>>(ws, N, C, LOSS_BLOCKS_PER_IMAGE, kind, eps, img_part);
    seg_loss_merge2_kernel<<<1, 64, 0, st>>>(ws, N, C, label_smoothing, class_weights, img_part, losses);
    YDL_LAUNCH_CHECK();
    return 0;
}

extern "C" int ydl_seg_loss_rep_bwd(const float* plow, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                                    const int64_t* target, const float* class_weights, int kind, float label_smoothing, float eps,
                                    int N, int C, int H, int W, int rep_h, int rep_w, const float* ws, const float* dloss,
                                    float* dlow, void* stream) {
    (void)kind; (void)eps;
    YDL_CHECK(plow && target && ws && dlow, "null pointer");
    YDL_CHECK(C >= 1 && C <= MAXC && rep_h >= 1 && rep_w >= 1, "bad sizes (C <= 32)");
    hipStream_t st = (hipStream_t)stream;
    long long HW = (long long)H * W;
    int bx = (int)((HW + 255) / 256);
    if (bx > 1024) bx = 1024;
    dim3 grid(bx, N);
    if (C <= 16)
        seg_loss_rep_bwd_kernel<16><<<grid, 256, 0, st>>>(plow, sn, sc, sh, sw, target, class_weights, C, H, W, rep_h, rep_w,
                                                          label_smoothing, ws, N, dloss, dlow);
    else
        seg_loss_rep_bwd_kernel<32><<<grid, 256, 0, st>>>(plow, sn, sc, sh, sw, target, class_weights, C, H, W, rep_h, rep_w,
                                                          label_smoothing, ws, N, dloss, dlow);
    YDL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// argmax + confusion matrix (rows = target, cols = prediction; pixels whose target == ignore_index dropped)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void confusion_kernel(const float* __restrict__ pred, long long sn, long long sc, long long sh,
                                                        long long sw, const int64_t* __restrict__ target, int N, int C, int H, int W,
                                                        int ignore, unsigned long long* __restrict__ matrix) {
    __shared__ unsigned int hist[MAXC * MAXC];
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const long long total = (long long)N * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int w = (int)(i % W);
        long long t2 = i / W;
        int h = (int)(t2 % H);
        int n = (int)(t2 / H);
        long long t = target[i];
        if (t < 0 || t >= C || t == ignore) continue;
        const float* pp = pred + n * sn + h * sh + w * sw;
        float best = pp[0];
        int bi = 0;
        for (int c = 1; c < C; ++c) {
            float v = pp[c * sc];
            if (v > best) { best = v; bi = c; }     // first maximum wins, like torch.argmax
        }
        atomicAdd(&hist[(int)t * C + bi], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += blockDim.x)
        if (hist[i]) atomicAdd(&matrix[i], (unsigned long long)hist[i]);
}

extern "C" int ydl_confusion_matrix(const float* pred, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                                    const int64_t* target, int N, int C, int H, int W, int ignore_index,
                                    int64_t* matrix, void* stream) {
    YDL_CHECK(pred && target && matrix && C >= 1 && C <= MAXC, "bad arguments (C <= 32)");
    long long total = (long long)N * H * W;
    int grid = (int)((total + 255) / 256);
    if (grid > 1024) grid = 1024;
    confusion_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(pred, sn, sc, sh, sw, target, N, C, H, W, ignore_index,
                                                             (unsigned long long*)matrix);
    YDL_LAUNCH_CHECK();
    return 0;
}
