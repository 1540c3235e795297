// Train-mode BatchNorm + activation (+ residual) for NHWC activations: statistics finalize, fused apply,
// and the two-phase backward.  All kernels are HBM-bound streaming passes with 16-byte accesses.
#include "common.h"

// ------------------------------------------------------------------------------------------------------
// finalize: merge of the per-block (sum, M2) partials written by the conv epilogue.
// partials: [nblocks][2][ldp]; block b covered n_b = min(block_m, count - b*block_m) pixels.
// ------------------------------------------------------------------------------------------------------
// level-1 pre-merge for very long partial lists (> 1024 rows: the 160x160 and 320x320 layers of the tiled kernel): chunks of
// MERGE_CHUNK rows are Chan-merged (two-pass, double) by grid.y CTAs into one (sum, M2) row each
#define MERGE_CHUNK 64

// merges blocks [b0, b1) for channel c; every thread of the CTA must call it (uses LDS + barriers).
// returns (sum, M2, n) on slice 0 threads.
__device__ __forceinline__ void chan_merge(const float* __restrict__ part, int b0, int b1, int block_m, long long count,
                                           int c, bool cvalid, int ldp, double& tot_o, double& m2_o, double& n_o) {
    __shared__ double sh[8][33];
    __shared__ double smean[32];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    double s = 0.0;
    if (cvalid)
        for (int b = b0 + sl; b < b1; b += 8) s += (double)part[(size_t)b * 2 * ldp + c];
    sh[sl][cl] = s;
    __syncthreads();
    long long n_lo = (long long)b0 * block_m, n_hi = (long long)b1 * block_m;
    if (n_hi > count) n_hi = count;
    const double ntot = (double)(n_hi - n_lo);
    if (sl == 0) {
        double tot = 0.0;
        for (int i = 0; i < 8; ++i) tot += sh[i][cl];
        smean[cl] = tot / ntot;
        tot_o = tot;
    }
    __syncthreads();
    const double mu = smean[cl];
    double m2 = 0.0;
    if (cvalid)
        for (int b = b0 + sl; b < b1; b += 8) {
            long long nb = count - (long long)b * block_m;
            if (nb > block_m) nb = block_m;
            double sb = (double)part[(size_t)b * 2 * ldp + c];
            double d = sb / (double)nb - mu;
            m2 += (double)part[(size_t)b * 2 * ldp + ldp + c] + (double)nb * d * d;
        }
    __syncthreads();
    sh[sl][cl] = m2;
    __syncthreads();
    if (sl == 0) {
        double t2 = 0.0;
        for (int i = 0; i < 8; ++i) t2 += sh[i][cl];
        m2_o = t2;
        n_o = ntot;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void bn_merge_level1_kernel(const float* __restrict__ part, int nblocks, int block_m,
                                                              long long count, int C, int ldp, float* __restrict__ part2) {
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    const int b0 = blockIdx.y * MERGE_CHUNK;
    const int b1 = min(nblocks, b0 + MERGE_CHUNK);
    double tot = 0, m2 = 0, n = 0;
    chan_merge(part, b0, b1, block_m, count, c, c < C, ldp, tot, m2, n);
    if ((threadIdx.x >> 5) == 0 && c < C) {
        part2[(size_t)blockIdx.y * 2 * ldp + c] = (float)tot;
        part2[(size_t)blockIdx.y * 2 * ldp + ldp + c] = (float)m2;
    }
}

// One-pass merge: with S = sum_b s_b and Q = sum_b (M2_b + s_b^2 / n_b), M2 = Q - S^2 / N (all in double: the
// subtraction loses log2(mean^2/var) of 53 bits).  8 channels x 32 slices per CTA; a slice walks its rows eight at a
// time with clamped (always valid) addresses and 0/1 weights, so the loads of a batch are independent and unpredicated.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nblocks, int block_m,
                                                          long long count, int C, int ldp,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float momentum, float* running_mean, float* running_var,
                                                          float* mean_o, float* invstd_o, float* scale_o, float* shift_o, int rep) {
    constexpr int CH = 8, NS = 256 / CH, B = 8;      // 8 channels x 32 row slices per CTA, 8 rows per batch
    __shared__ double sh[2][NS][CH + 1];
    const int cl = threadIdx.x % CH, sl = threadIdx.x / CH;
    const int c = blockIdx.x * CH + cl;
    const int cc = c < C ? c : C - 1;
    // coefficient loads issued before the merge so that their latency hides behind it
    const float g = gamma ? gamma[cc] : 1.f, bt = beta ? beta[cc] : 0.f;
    const float rm = running_mean ? running_mean[cc] : 0.f, rv = running_var ? running_var[cc] : 0.f;
    const double inv_bm = 1.0 / (double)block_m;
    const long long last_n = count - (long long)(nblocks - 1) * block_m;
    const double inv_last = 1.0 / (double)last_n;
    double S = 0.0, Q = 0.0;
    for (int base = sl; base < nblocks; base += NS * B) {
        float sv[B], qv[B];
        int bi[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const int b = base + NS * u;
            bi[u] = b < nblocks ? b : nblocks - 1;
            sv[u] = part[(size_t)bi[u] * 2 * ldp + cc];
            qv[u] = part[(size_t)bi[u] * 2 * ldp + ldp + cc];
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const double w = (base + NS * u) < nblocks ? 1.0 : 0.0;
            const double s = (double)sv[u];
            const double inb = bi[u] == nblocks - 1 ? inv_last : inv_bm;
            S += w * s;
            Q += w * ((double)qv[u] + s * s * inb);
        }
    }
    sh[0][sl][cl] = S;
    sh[1][sl][cl] = Q;
    __syncthreads();
    if (sl == 0 && c < C) {
        double tot = 0.0, q = 0.0;
        for (int i = 0; i < NS; ++i) { tot += sh[0][i][cl]; q += sh[1][i][cl]; }
        double mu = tot / (double)count;
        double m2 = q - tot * mu;
        if (m2 < 0.0) m2 = 0.0;
        double var = m2 / (double)count;            // biased (normalisation)
        float meanf = (float)mu;
        float invstd = (float)(1.0 / sqrt(var + (double)eps));
        mean_o[c] = meanf;
        invstd_o[c] = invstd;
        float sc = g * invstd;
        scale_o[c] = sc;
        shift_o[c] = bt - meanf * sc;
        if (running_mean) {
            // the logical tensor is the stored one replicated `rep` times: M2 and the count scale by rep
            double nl = (double)count * (double)rep;
            double unb = nl > 1.0 ? m2 * (double)rep / (nl - 1.0) : var;
            running_mean[c] = (1.f - momentum) * rm + momentum * meanf;
            running_var[c] = (1.f - momentum) * rv + momentum * (float)unb;
        }
    }
}

extern "C" int ydl_bn_finalize(const float* stats_ws, int nblocks, int block_m, int64_t count, int C,
                               const float* gamma, const float* beta, float eps, float momentum,
                               float* running_mean, float* running_var, float* mean, float* invstd,
                               float* scale, float* shift, int replication, void* stream) {
    YDL_CHECK(stats_ws && mean && invstd && scale && shift, "null pointer");
    YDL_CHECK(nblocks > 0 && block_m > 0 && count > 0 && C > 0 && replication >= 1, "bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int ldp = round_up(C, 8);
    const float* src = stats_ws;
    ydl_note_kernel(3, nblocks > 1024 ? "bn_finalize<two-level>" : "bn_finalize<one-level>");
    if (nblocks > 1024) {
        // level 1 writes behind the level-0 partials (the workspace query reserves the room)
        int nchunks = (nblocks + MERGE_CHUNK - 1) / MERGE_CHUNK;
        float* part2 = const_cast<float*>(stats_ws) + (size_t)nblocks * 2 * ldp;
        bn_merge_level1_kernel<<<dim3((C + 31) / 32, nchunks), 256, 0, st>>>(stats_ws, nblocks, block_m, (long long)count, C, ldp, part2);
        src = part2;
        nblocks = nchunks;
        block_m *= MERGE_CHUNK;
    }
    bn_finalize_kernel<<<(C + 7) / 8, 256, 0, st>>>(src, nblocks, block_m, (long long)count, C, ldp, gamma, beta, eps,
                                                    momentum, running_mean, running_var, mean, invstd, scale, shift, replication);
    YDL_LAUNCH_CHECK();
    return 0;
}

__global__ void bn_eval_coeffs_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, float* scale, float* shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        float sc = gamma[c] / sqrtf(rv[c] + eps);
        scale[c] = sc;
        shift[c] = beta[c] - rm[c] * sc;
    }
}
extern "C" int ydl_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* scale, float* shift, void* stream) {
    YDL_CHECK(C > 0 && gamma && beta && running_mean && running_var && scale && shift, "bad arguments");
    bn_eval_coeffs_kernel<<<(C + 255) / 256, 256, 0, (hipStream_t)stream>>>(C, gamma, beta, running_mean, running_var, eps, scale, shift);
    YDL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// Streaming kernels: a thread owns ONE 16-byte channel chunk for the whole kernel (its scale/shift/mean/invstd sit
// in registers) and walks pixels with a 32-bit stride; a CTA covers R = 256/cpb pixels per iteration, where
// cpb = min(chunks per pixel, 256).  No 64-bit divisions, no per-element coefficient loads.
// ------------------------------------------------------------------------------------------------------
#ifndef BN_APPLY_REVERSE
#define BN_APPLY_REVERSE 1
#endif
struct Lay { int cpb, R, cq, pl; bool live; int c; };
template <int V>
__device__ __forceinline__ Lay make_lay(int Cp) {
    Lay L;
    const int cpp = Cp / V;
    L.cpb = cpp < 256 ? cpp : 256;
    L.R = 256 / L.cpb;
    L.cq = threadIdx.x % L.cpb;
    L.pl = threadIdx.x / L.cpb;
    const int chunk = blockIdx.y * 256 + L.cq;
    L.live = L.pl < L.R && chunk < cpp;
    L.c = chunk * V;
    return L;
}
static inline dim3 lay_grid(long long npix, int Cp, int V, int max_x) {
    int cpp = Cp / V;
    int cpb = cpp < 256 ? cpp : 256;
    int R = 256 / cpb;
    long long gx = (npix + R - 1) / R;
    if (gx > max_x) gx = max_x;
    if (gx < 1) gx = 1;
    return dim3((unsigned)gx, (unsigned)((cpp + 255) / 256));
}

// apply: out = act(y*scale + shift) (+res).
// scale/shift arrays must be readable up to Cp (padded entries = 0 => padded channels stay 0 for SiLU/none).
template <typename T, int ACT, int RES>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const T* __restrict__ res, int ldr,
                                                         T* __restrict__ out, int ldo,
                                                         long long npix, int Cp) {
    constexpr int V = ET<T>::V;
    const Lay L = make_lay<V>(Cp);
    if (!L.live) return;
    float sc[V], sf[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { sc[e] = scale[L.c + e]; sf[e] = shift[L.c + e]; }
    const long long stride = (long long)gridDim.x * L.R;
    for (long long pix = (long long)blockIdx.x * L.R + L.pl; pix < npix; pix += stride) {
        float v[V], r[V];
        unpack16<T>(*(const uint4*)(y + pix * ldy + L.c), v);
        if (RES != YDL_RES_NONE) unpack16<T>(*(const uint4*)(res + pix * ldr + L.c), r);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float z = v[e] * sc[e] + sf[e];
            if (RES == YDL_RES_BEFORE_ACT) z += r[e];
            float o = ACT == YDL_ACT_SILU ? silu_f(z) : (ACT == YDL_ACT_RELU ? fmaxf(z, 0.f) : z);
            if (RES == YDL_RES_AFTER_ACT) o += r[e];
            v[e] = o;
        }
        *(uint4*)(out + pix * ldo + L.c) = pack16<T>(v);
    }
}

extern "C" int ydl_bn_act_fwd(int dtype, const void* y, int ldy, const float* scale, const float* shift,
                              const void* res, int ldr, int res_mode, int act, void* out, int ldo,
                              int64_t npix, int Cp, void* stream) {
    const int V = dtype == YDL_F32 ? 4 : 8;
    YDL_CHECK(y && out && scale && shift, "null pointer");
    YDL_CHECK(Cp > 0 && Cp % V == 0 && ldy >= Cp && ldo >= Cp, "Cp must be a chunk multiple covered by the strides");
    YDL_CHECK(res_mode == YDL_RES_NONE || (res != nullptr && ldr >= Cp), "residual requested but missing");
    YDL_CHECK(aligned16(y) && aligned16(out) && (res == nullptr || aligned16(res)), "16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid = lay_grid(npix, Cp, V, 256 * 8);
    YDL_CHECK(act == YDL_ACT_NONE || act == YDL_ACT_SILU || act == YDL_ACT_RELU, "unknown activation");
    YDL_CHECK(res_mode == YDL_RES_NONE || res_mode == YDL_RES_BEFORE_ACT || res_mode == YDL_RES_AFTER_ACT, "unknown residual mode");
    // activation and residual mode are compile-time: a run-time switch inside the unrolled element loop made the compiler emit a
    // scalar branch per element
#define YDL_FWD_LAUNCH(T, A, R) bn_act_fwd_kernel<T, A, R><<<grid, 256, 0, st>>>((const T*)y, ldy, scale, shift, (const T*)res, ldr, (T*)out, ldo, npix, Cp)
#define YDL_FWD_RES(T, A)                                                                  \
    do {                                                                                   \
        if (res_mode == YDL_RES_NONE) YDL_FWD_LAUNCH(T, A, YDL_RES_NONE);                  \
        else if (res_mode == YDL_RES_BEFORE_ACT) YDL_FWD_LAUNCH(T, A, YDL_RES_BEFORE_ACT); \
        else YDL_FWD_LAUNCH(T, A, YDL_RES_AFTER_ACT);                                      \
    } while (0)
#define YDL_FWD_ACT(T)                                                 \
    do {                                                               \
        if (act == YDL_ACT_SILU) YDL_FWD_RES(T, YDL_ACT_SILU);         \
        else if (act == YDL_ACT_RELU) YDL_FWD_RES(T, YDL_ACT_RELU);    \
        else YDL_FWD_RES(T, YDL_ACT_NONE);                             \
    } while (0)
    if (dtype == YDL_F32) YDL_FWD_ACT(float);
    else YDL_FWD_ACT(bf16_t);
    YDL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// backward.  dz = dout * act'(z);  dbeta = sum dz;  dgamma = sum dz*xhat;
//            dy = gamma*invstd * (dz - dbeta/M - xhat*dgamma/M)
// phase 1: per-CTA partial (dbeta, dgamma), fixed pixel->CTA assignment (deterministic) ; phase 2: merge in double ;
// phase 3: streaming apply.   ws layout (floats): [nblk][2][Cp] partials, then [2][Cp] merged sums.
// ------------------------------------------------------------------------------------------------------
#define BWD_MAX_PARTIALS 1024

static inline int bwd_nblk(long long npix, int Cp, int V) {
    return (int)lay_grid(npix, Cp, V, BWD_MAX_PARTIALS).x;
}

template <typename T, int ACT>
__device__ __forceinline__ void dz_xhat_q(const uint4& yq, const uint4& dq, const uint4& oq, const float* sc, const float* sf,
                                          const float* mu, const float* is, float* dz, float* xh) {
    constexpr int V = ET<T>::V;
    float yv[V], dv[V], ov[V];
    unpack16<T>(yq, yv);
    unpack16<T>(dq, dv);
    if (ACT == YDL_ACT_RELU) unpack16<T>(oq, ov);
#pragma unroll
    for (int e = 0; e < V; ++e) {
        float z = yv[e] * sc[e] + sf[e];
        float d = dv[e];
        if (ACT == YDL_ACT_SILU) {
            float sg = sigmoid_f(z);
            d *= sg * (1.f + z * (1.f - sg));
        } else if (ACT == YDL_ACT_RELU) {
            d = ov[e] > 0.f ? d : 0.f;
        }
        dz[e] = d;
        xh[e] = (yv[e] - mu[e]) * is[e];
    }
}

template <typename T, int ACT>
__device__ __forceinline__ void dz_xhat(const T* y, const T* dout, const T* out, long long pix, int ldy, int lddo, int ldo,
                                        int c, const float* sc, const float* sf, const float* mu, const float* is,
                                        float* dz, float* xh, float* dv) {
    constexpr int V = ET<T>::V;
    float yv[V], ov[V];
    unpack16<T>(*(const uint4*)(y + pix * ldy + c), yv);
    unpack16<T>(*(const uint4*)(dout + pix * lddo + c), dv);
    if (ACT == YDL_ACT_RELU) unpack16<T>(*(const uint4*)(out + pix * ldo + c), ov);
#pragma unroll
    for (int e = 0; e < V; ++e) {
        float z = yv[e] * sc[e] + sf[e];
        float d = dv[e];
        if (ACT == YDL_ACT_SILU) {
            float sg = sigmoid_f(z);
            d *= sg * (1.f + z * (1.f - sg));
        } else if (ACT == YDL_ACT_RELU) {
            d = ov[e] > 0.f ? d : 0.f;
        }
        dz[e] = d;
        xh[e] = (yv[e] - mu[e]) * is[e];
    }
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ y, int ldy, const T* __restrict__ dout, int lddo,
                                                            const T* __restrict__ out, int ldo,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            float* __restrict__ part, long long npix, int Cp) {
    constexpr int V = ET<T>::V;
    const Lay L = make_lay<V>(Cp);
    float sb[V], sg[V], sc[V], sf[V], mu[V], is[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { sb[e] = 0.f; sg[e] = 0.f; }
    if (L.live) {
#pragma unroll
        for (int e = 0; e < V; ++e) { sc[e] = scale[L.c + e]; sf[e] = shift[L.c + e]; mu[e] = mean[L.c + e]; is[e] = invstd[L.c + e]; }
        const long long stride = (long long)gridDim.x * L.R;
        long long pix = (long long)blockIdx.x * L.R + L.pl;
        // two pixels per iteration: four independent 16-byte loads in flight per thread (the grid is capped at 1024 partial rows,
        // so memory-level parallelism has to come from inside the thread)
        for (; pix + stride < npix; pix += 2 * stride) {
            const uint4 y0 = *(const uint4*)(y + pix * ldy + L.c), d0 = *(const uint4*)(dout + pix * lddo + L.c);
            const uint4 y1 = *(const uint4*)(y + (pix + stride) * ldy + L.c), d1 = *(const uint4*)(dout + (pix + stride) * lddo + L.c);
            uint4 o0 = make_uint4(0, 0, 0, 0), o1 = o0;
            if (ACT == YDL_ACT_RELU) { o0 = *(const uint4*)(out + pix * ldo + L.c); o1 = *(const uint4*)(out + (pix + stride) * ldo + L.c); }
            float dz[V], xh[V];
            dz_xhat_q<T, ACT>(y0, d0, o0, sc, sf, mu, is, dz, xh);
#pragma unroll
            for (int e = 0; e < V; ++e) { sb[e] += dz[e]; sg[e] += dz[e] * xh[e]; }
            dz_xhat_q<T, ACT>(y1, d1, o1, sc, sf, mu, is, dz, xh);
#pragma unroll
            for (int e = 0; e < V; ++e) { sb[e] += dz[e]; sg[e] += dz[e] * xh[e]; }
        }
        for (; pix < npix; pix += stride) {
            float dz[V], xh[V], dvr[V];
            dz_xhat<T, ACT>(y, dout, out, pix, ldy, lddo, ldo, L.c, sc, sf, mu, is, dz, xh, dvr);
#pragma unroll
            for (int e = 0; e < V; ++e) { sb[e] += dz[e]; sg[e] += dz[e] * xh[e]; }
        }
    }
    // reduce over the CTA's pixel lanes through LDS
    __shared__ float red[256 * 2 * 8];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        red[(threadIdx.x * 2 + 0) * V + e] = sb[e];
        red[(threadIdx.x * 2 + 1) * V + e] = sg[e];
    }
    __syncthreads();
    if (threadIdx.x < L.cpb && L.live) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float a = 0.f, b = 0.f;
            for (int l = 0; l < L.R; ++l) {
                int tt = l * L.cpb + threadIdx.x;
                a += red[(tt * 2 + 0) * V + e];
                b += red[(tt * 2 + 1) * V + e];
            }
            float* dst = part + (size_t)blockIdx.x * 2 * Cp;
            dst[L.c + e] = a;
            dst[Cp + L.c + e] = b;
        }
    }
}

// 8 channels x 32 slices per CTA; four independent accumulators keep the partial loads in flight
__global__ __launch_bounds__(256) void bn_bwd_merge_kernel(const float* __restrict__ part, int nblk, int Cp, int C,
                                                           float* __restrict__ sums, float* dgamma, float* dbeta, int accumulate) {
    __shared__ double sh[2][32][9];
    const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    const int cc = c < Cp ? c : Cp - 1;
    const float g0 = (accumulate && dgamma && c < C) ? dgamma[c] : 0.f;     // issued early: hidden behind the merge
    const float b0 = (accumulate && dbeta && c < C) ? dbeta[c] : 0.f;
    double a = 0, b = 0;
    for (int base = sl; base < nblk; base += 32 * 4) {
        float av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                   // clamped row index: unpredicated, independent loads
            const int i = base + 32 * u;
            const int ii = i < nblk ? i : nblk - 1;
            av[u] = part[(size_t)ii * 2 * Cp + cc];
            bv[u] = part[(size_t)ii * 2 * Cp + Cp + cc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double w = (base + 32 * u) < nblk ? 1.0 : 0.0;
            a += w * (double)av[u];
            b += w * (double)bv[u];
        }
    }
    sh[0][sl][cl] = a;
    sh[1][sl][cl] = b;
    __syncthreads();
    if (sl == 0 && c < Cp) {
        double ta = 0.0, tb = 0.0;
        for (int i = 0; i < 32; ++i) { ta += sh[0][i][cl]; tb += sh[1][i][cl]; }
        sums[c] = (float)ta;          // dbeta
        sums[Cp + c] = (float)tb;     // dgamma
        if (c < C) {
            if (dbeta) dbeta[c] = b0 + (float)ta;
            if (dgamma) dgamma[c] = g0 + (float)tb;
        }
    }
}

// RESM: 0 no residual gradient, 1 dres (+)= dout (residual added after the activation), 2 dres (+)= dz (added before it)
template <typename T, int ACT, int RESM>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ y, int ldy, const T* __restrict__ dout, int lddo,
                                                           const T* __restrict__ out, int ldo,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ sums, T* __restrict__ dy, int lddy,
                                                           T* __restrict__ dres, int lddr, int dres_acc, long long npix, int Cp) {
    constexpr int V = ET<T>::V;
    const Lay L = make_lay<V>(Cp);
    if (!L.live) return;
    const float invM = 1.0f / (float)npix;
    float sc[V], sf[V], mu[V], is[V], kb[V], kg[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        sc[e] = scale[L.c + e]; sf[e] = shift[L.c + e]; mu[e] = mean[L.c + e]; is[e] = invstd[L.c + e];
        kb[e] = sums[L.c + e] * invM; kg[e] = sums[Cp + L.c + e] * invM;
    }
    const long long stride = (long long)gridDim.x * L.R;
    // the second pass walks the tensor BACKWARDS: what the reduce pass read last is the likeliest to still sit in L2 / the
    // Infinity Cache (BN_APPLY_REVERSE=0: same order as the reduce pass, for A/B timing)
    const long long pfirst = (long long)blockIdx.x * L.R + L.pl;
    const long long plast = pfirst < npix ? pfirst + (npix - 1 - pfirst) / stride * stride : pfirst - stride;
    for (long long pix = BN_APPLY_REVERSE ? plast : pfirst; BN_APPLY_REVERSE ? pix >= pfirst : pix < npix;
         pix += BN_APPLY_REVERSE ? -stride : stride) {
        float dz[V], xh[V], o[V], dv[V];
        dz_xhat<T, ACT>(y, dout, out, pix, ldy, lddo, ldo, L.c, sc, sf, mu, is, dz, xh, dv);
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = sc[e] * (dz[e] - kb[e] - xh[e] * kg[e]);
        *(uint4*)(dy + pix * lddy + L.c) = pack16<T>(o);
        if (RESM != 0) {                                  // the residual branch's gradient in the same pass (no separate copy kernel)
            T* rp = dres + pix * lddr + L.c;
            float r[V];
#pragma unroll
            for (int e = 0; e < V; ++e) r[e] = RESM == 1 ? dv[e] : dz[e];
            if (dres_acc) {
                float old[V];
                unpack16<T>(*(const uint4*)rp, old);
#pragma unroll
                for (int e = 0; e < V; ++e) r[e] += old[e];
            }
            *(uint4*)rp = pack16<T>(r);
        }
    }
}

extern "C" int64_t ydl_bn_bwd_ws_bytes(int64_t npix, int Cp) {
    (void)npix;
    return ((int64_t)BWD_MAX_PARTIALS * 2 * Cp + 2 * Cp) * (int64_t)sizeof(float);
}

extern "C" int ydl_bn_act_bwd(int dtype, const void* y, int ldy, const void* dout, int lddo, const void* out, int ldo,
                              const float* gamma, const float* mean, const float* invstd, const float* scale, const float* shift,
                              int res_mode, int act, void* dy, int lddy, void* dres, int lddr,
                              float* dgamma, float* dbeta, int accumulate_param_grads,
                              float* ws, int64_t npix, int C, int Cp, void* stream) {
    (void)gamma;
    const int dres_acc = (res_mode & YDL_RES_GRAD_ACCUMULATE) ? 1 : 0;
    const int rmode = res_mode & 15;
    YDL_CHECK(rmode == YDL_RES_NONE || rmode == YDL_RES_AFTER_ACT || rmode == YDL_RES_BEFORE_ACT, "unknown residual mode");
    const int resm = dres == nullptr ? 0 : (rmode == YDL_RES_AFTER_ACT ? 1 : 2);     // dres without a mode: the masked gradient
    YDL_CHECK(dres == nullptr || (lddr >= Cp && aligned16(dres)), "dres must be 16-byte aligned with a stride covering Cp");
    const int V = dtype == YDL_F32 ? 4 : 8;
    YDL_CHECK(y && dout && dy && mean && invstd && scale && shift && ws, "null pointer");
    YDL_CHECK(act != YDL_ACT_RELU || out != nullptr, "RELU backward needs the saved output");
    YDL_CHECK(Cp > 0 && Cp % V == 0 && C <= Cp && ldy >= Cp && lddo >= Cp && lddy >= Cp, "bad channel geometry");
    YDL_CHECK(aligned16(y) && aligned16(dout) && aligned16(dy), "16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    dim3 g1 = lay_grid(npix, Cp, V, BWD_MAX_PARTIALS);
    const int nblk = (int)g1.x;
    float* part = ws;
    float* sums = ws + (size_t)BWD_MAX_PARTIALS * 2 * Cp;
    dim3 g3 = lay_grid(npix, Cp, V, 256 * 8);
    YDL_CHECK(act == YDL_ACT_NONE || act == YDL_ACT_SILU || act == YDL_ACT_RELU, "unknown activation");
#define YDL_BWD_LAUNCH(T, A)                                                                                                          \
    do {                                                                                                                              \
        bn_bwd_reduce_kernel<T, A><<<g1, 256, 0, st>>>((const T*)y, ldy, (const T*)dout, lddo, (const T*)out, ldo, scale, shift,     \
                                                       mean, invstd, part, npix, Cp);                                                \
        bn_bwd_merge_kernel<<<(Cp + 7) / 8, 256, 0, st>>>(part, nblk, Cp, C, sums, dgamma, dbeta, accumulate_param_grads);            \
        if (resm == 0)                                                                                                                \
            bn_bwd_apply_kernel<T, A, 0><<<g3, 256, 0, st>>>((const T*)y, ldy, (const T*)dout, lddo, (const T*)out, ldo, scale,      \
                                                             shift, mean, invstd, sums, (T*)dy, lddy, (T*)dres, lddr, dres_acc, npix, Cp); \
        else if (resm == 1)                                                                                                           \
            bn_bwd_apply_kernel<T, A, 1><<<g3, 256, 0, st>>>((const T*)y, ldy, (const T*)dout, lddo, (const T*)out, ldo, scale,      \
                                                             shift, mean, invstd, sums, (T*)dy, lddy, (T*)dres, lddr, dres_acc, npix, Cp); \
        else                                                                                                                          \
            bn_bwd_apply_kernel<T, A, 2><<<g3, 256, 0, st>>>((const T*)y, ldy, (const T*)dout, lddo, (const T*)out, ldo, scale,      \
                                                             shift, mean, invstd, sums, (T*)dy, lddy, (T*)dres, lddr, dres_acc, npix, Cp); \
    } while (0)
#define YDL_BWD_ACT(T)                                                                                                                \
    do {                                                                                                                              \
        if (act == YDL_ACT_SILU) YDL_BWD_LAUNCH(T, YDL_ACT_SILU);                                                                     \
        else if (act == YDL_ACT_RELU) YDL_BWD_LAUNCH(T, YDL_ACT_RELU);                                                                \
        else YDL_BWD_LAUNCH(T, YDL_ACT_NONE);                                                                                         \
    } while (0)
    if (dtype == YDL_F32) YDL_BWD_ACT(float);
    else YDL_BWD_ACT(bf16_t);
    YDL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// Throughput-mode forms on replica sums (ydl.h: ydl_conv_fwd_sums / ydl_bn_act_fwd_sums / ydl_bn_act_bwd_sums): the
// finalize and merge launches disappear — 62 launches of 5-9 us per step on BASELINE config 2 — because every consumer thread
// adds the YDL_BN_REPLICAS partial rows of its own channel chunk itself (L2-resident, a few hundred bytes).
// ------------------------------------------------------------------------------------------------------
#ifndef BN_FWD_UNROLL
#define BN_FWD_UNROLL 4
#endif
template <typename T, int ACT, int RES>
__global__ __launch_bounds__(256) void bn_act_fwd_sums_kernel(const T* __restrict__ y, int ldy, const float* __restrict__ sums, int sums_ld,
                                                              long long count, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, float momentum,
                                                              float* running_mean, float* running_var, float* mean_o, float* invstd_o,
                                                              float* scale_o, float* shift_o, int rep, const T* __restrict__ res, int ldr,
                                                              T* __restrict__ out, int ldo, long long npix, int C, int Cp) {
    constexpr int V = ET<T>::V;
    const Lay L = make_lay<V>(Cp);
    // coefficients of the CTA's channels: ONE channel per thread (replica rows summed in double), shared through LDS — deriving
    // them per thread for its whole chunk cost more than the streaming loop itself
    __shared__ float s_sc[256 * 8], s_sf[256 * 8];
    {
        const int nch = L.cpb * V;
        const int cbase = blockIdx.y * 256 * V;
        const double inv_n = 1.0 / (double)count;
        for (int j = threadIdx.x; j < nch; j += 256) {
            const int c = cbase + j;
            float scv = 0.f, sfv = 0.f;
            if (c < Cp) {
                float meanf = 0.f, invstd = 0.f;
                double m2 = 0.0, var = 0.0;
                if (c < C) {
                    double s1 = 0.0, s2 = 0.0;
#pragma unroll
                    for (int r = 0; r < YDL_BN_REPLICAS; ++r) {
                        s1 += (double)sums[(size_t)(2 * r) * sums_ld + c];
                        s2 += (double)sums[(size_t)(2 * r + 1) * sums_ld + c];
                    }
                    const double mu = s1 * inv_n;
                    m2 = s2 - s1 * mu;
                    if (m2 < 0.0) m2 = 0.0;
                    var = m2 * inv_n;
                    meanf = (float)mu;
                    invstd = (float)(1.0 / sqrt(var + (double)eps));
                    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
                    scv = g * invstd;
                    sfv = bt - meanf * scv;
                }
                if (blockIdx.x == 0) {
                    mean_o[c] = meanf; invstd_o[c] = invstd; scale_o[c] = scv; shift_o[c] = sfv;
                    if (running_mean && c < C) {
                        const double nl = (double)count * (double)rep;
                        const double unb = nl > 1.0 ? m2 * (double)rep / (nl - 1.0) : var;
                        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * meanf;
                        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
                    }
                }
            }
            s_sc[j] = scv; s_sf[j] = sfv;
        }
    }
    __syncthreads();
    if (!L.live) return;
    float sc[V], sf[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { sc[e] = s_sc[L.cq * V + e]; sf[e] = s_sf[L.cq * V + e]; }
    const long long stride = (long long)gridDim.x * L.R;
    // BN_FWD_UNROLL pixels in flight per thread, and a grid sized (launcher) so that a thread HAS that many: on the 40^2 / 20^2 layers
    // the former 2048-CTA grid gave every CTA one pixel row after a ~2 us coefficient prologue (8-9 us for 13 MB of traffic)
    constexpr int U = BN_FWD_UNROLL;
    for (long long pix0 = (long long)blockIdx.x * L.R + L.pl; pix0 < npix; pix0 += stride * U) {
        uint4 raw[U], rraw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long pix = pix0 + u * stride;
            const long long pc = pix < npix ? pix : pix0;          // (beyond the end: re-read this thread's first pixel, store nothing)
            raw[u] = *(const uint4*)(y + pc * ldy + L.c);
            if (RES != YDL_RES_NONE) rraw[u] = *(const uint4*)(res + pc * ldr + L.c);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long pix = pix0 + u * stride;
            float v[V], r[V];
            unpack16<T>(raw[u], v);
            if (RES != YDL_RES_NONE) unpack16<T>(rraw[u], r);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float z = v[e] * sc[e] + sf[e];
                if (RES == YDL_RES_BEFORE_ACT) z += r[e];
                float o = ACT == YDL_ACT_SILU ? silu_f(z) : (ACT == YDL_ACT_RELU ? fmaxf(z, 0.f) : z);
                if (RES == YDL_RES_AFTER_ACT) o += r[e];
                v[e] = o;
            }
            if (pix < npix) *(uint4*)(out + pix * ldo + L.c) = pack16<T>(v);
        }
    }
}

extern "C" int ydl_bn_act_fwd_sums(int dtype, const void* y, int ldy, const float* sums, int sums_ld, int64_t count,
                                   const float* gamma, const float* beta, float eps, float momentum,
                                   float* running_mean, float* running_var, float* mean, float* invstd, float* scale, float* shift,
                                   int replication, const void* res, int ldr, int res_mode, int act, void* out, int ldo,
                                   int64_t npix, int C, int Cp, void* stream) {
    const int V = dtype == YDL_F32 ? 4 : 8;
    YDL_CHECK(dtype == YDL_F32 || dtype == YDL_BF16, "bad dtype");
    YDL_CHECK(y && out && sums && mean && invstd && scale && shift, "null pointer");
    YDL_CHECK(count > 0 && replication >= 1 && C > 0 && C <= Cp, "bad sizes");
    YDL_CHECK(Cp % V == 0 && ldy >= Cp && ldo >= Cp && sums_ld >= Cp && sums_ld % 4 == 0, "Cp must be a chunk multiple covered by the strides");
    YDL_CHECK((running_mean == nullptr) == (running_var == nullptr), "running_mean and running_var come together");
    YDL_CHECK(res_mode == YDL_RES_NONE || (res != nullptr && ldr >= Cp), "residual requested but missing");
    YDL_CHECK(aligned16(y) && aligned16(out) && aligned16(sums) && (res == nullptr || aligned16(res)), "16-byte alignment");
    YDL_CHECK(act == YDL_ACT_NONE || act == YDL_ACT_SILU || act == YDL_ACT_RELU, "unknown activation");
    YDL_CHECK(res_mode == YDL_RES_NONE || res_mode == YDL_RES_BEFORE_ACT || res_mode == YDL_RES_AFTER_ACT, "unknown residual mode");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid = lay_grid(npix, Cp, V, 256 * 8);
    {
        // BN_FWD_UNROLL pixel rows per CTA pass: no more CTAs than there are passes (every CTA pays the coefficient prologue)
        static const int small = getenv("YDL_BN_SMALLGRID") ? atoi(getenv("YDL_BN_SMALLGRID")) : 1;
        const int cpp = Cp / V, cpb = cpp < 256 ? cpp : 256, R = 256 / cpb;
        const long long want = (npix + (long long)R * BN_FWD_UNROLL - 1) / ((long long)R * BN_FWD_UNROLL);
        if (small && want < (long long)grid.x) grid.x = (unsigned)(want < 1 ? 1 : want);
    }
#define YDL_FS_LAUNCH(T, A, R)                                                                                                      \
    bn_act_fwd_sums_kernel<T, A, R><<<grid, 256, 0, st>>>((const T*)y, ldy, sums, sums_ld, (long long)count, gamma, beta, eps, momentum, \
                                                          running_mean, running_var, mean, invstd, scale, shift, replication,         \
                                                          (const T*)res, ldr, (T*)out, ldo, npix, C, Cp)
#define YDL_FS_RES(T, A)                                                                  \
    do {                                                                                  \
        if (res_mode == YDL_RES_NONE) YDL_FS_LAUNCH(T, A, YDL_RES_NONE);                  \
        else if (res_mode == YDL_RES_BEFORE_ACT) YDL_FS_LAUNCH(T, A, YDL_RES_BEFORE_ACT); \
        else YDL_FS_LAUNCH(T, A, YDL_RES_AFTER_ACT);                                      \
    } while (0)
#define YDL_FS_ACT(T)                                                 \
    do {                                                              \
        if (act == YDL_ACT_SILU) YDL_FS_RES(T, YDL_ACT_SILU);         \
        else if (act == YDL_ACT_RELU) YDL_FS_RES(T, YDL_ACT_RELU);    \
        else YDL_FS_RES(T, YDL_ACT_NONE);                             \
    } while (0)
    if (dtype == YDL_F32) YDL_FS_ACT(float);
    else YDL_FS_ACT(bf16_t);
    YDL_LAUNCH_CHECK();
    return 0;
}

// reduce pass: as bn_bwd_reduce_kernel, but the CTA's per-channel (sum dz, sum dz*xhat) are ADDED to replica (blockIdx.x & 7)
// of sums; the final LDS pass is laid out one channel per thread so that a wave-instruction adds 256 contiguous bytes
template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_sums_kernel(const T* __restrict__ y, int ldy, const T* __restrict__ dout, int lddo,
                                                                 const T* __restrict__ out, int ldo,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 float* __restrict__ sums, long long npix, int Cp) {
    constexpr int V = ET<T>::V;
    const Lay L = make_lay<V>(Cp);
    float sb[V], sg[V], sc[V], sf[V], mu[V], is[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { sb[e] = 0.f; sg[e] = 0.f; }
    if (L.live) {
#pragma unroll
        for (int e = 0; e < V; ++e) { sc[e] = scale[L.c + e]; sf[e] = shift[L.c + e]; mu[e] = mean[L.c + e]; is[e] = invstd[L.c + e]; }
        const long long stride = (long long)gridDim.x * L.R;
        long long pix = (long long)blockIdx.x * L.R + L.pl;
        for (; pix + stride < npix; pix += 2 * stride) {
            const uint4 y0 = *(const uint4*)(y + pix * ldy + L.c), d0 = *(const uint4*)(dout + pix * lddo + L.c);
            const uint4 y1 = *(const uint4*)(y + (pix + stride) * ldy + L.c), d1 = *(const uint4*)(dout + (pix + stride) * lddo + L.c);
            uint4 o0 = make_uint4(0, 0, 0, 0), o1 = o0;
            if (ACT == YDL_ACT_RELU) { o0 = *(const uint4*)(out + pix * ldo + L.c); o1 = *(const uint4*)(out + (pix + stride) * ldo + L.c); }
            float dz[V], xh[V];
            dz_xhat_q<T, ACT>(y0, d0, o0, sc, sf, mu, is, dz, xh);
#pragma unroll
            for (int e = 0; e < V; ++e) { sb[e] += dz[e]; sg[e] += dz[e] * xh[e]; }
            dz_xhat_q<T, ACT>(y1, d1, o1, sc, sf, mu, is, dz, xh);
#pragma unroll
            for (int e = 0; e < V; ++e) { sb[e] += dz[e]; sg[e] += dz[e] * xh[e]; }
        }
        for (; pix < npix; pix += stride) {
            float dz[V], xh[V], dvr[V];
            dz_xhat<T, ACT>(y, dout, out, pix, ldy, lddo, ldo, L.c, sc, sf, mu, is, dz, xh, dvr);
#pragma unroll
            for (int e = 0; e < V; ++e) { sb[e] += dz[e]; sg[e] += dz[e] * xh[e]; }
        }
    }
    __shared__ float red[256 * 2 * 8];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        red[(threadIdx.x * 2 + 0) * V + e] = sb[e];
        red[(threadIdx.x * 2 + 1) * V + e] = sg[e];
    }
    __syncthreads();
    const int cpp = Cp / V;
    const int nch = L.cpb * V;                      // channels this CTA covers (from chunk blockIdx.y * 256)
    float* dst = sums + (size_t)(blockIdx.x & (YDL_BN_REPLICAS - 1)) * 2 * Cp;
    // narrow tensors (few chunks per pixel) have many pixel lanes per channel: G threads share one channel's lane sum
    // (16 channels x 128 lanes summed by 16 threads took longer than the streaming loop)
    const int G = nch >= 256 ? 1 : 256 / nch;
    __shared__ float part[256 * 2];
    float a = 0.f, b = 0.f;
    int j = threadIdx.x, grp = 0;
    if (G > 1) { j = threadIdx.x % nch; grp = threadIdx.x / nch; }
    if (G > 1) {
        if (grp < G) {
            const int cq = j / V, e = j - cq * V;
            for (int l = grp; l < L.R; l += G) {
                const int tt = l * L.cpb + cq;
                a += red[(tt * 2 + 0) * V + e];
                b += red[(tt * 2 + 1) * V + e];
            }
        }
        part[threadIdx.x * 2] = a;
        part[threadIdx.x * 2 + 1] = b;
        __syncthreads();
        if (threadIdx.x < nch) {
            a = 0.f; b = 0.f;
            for (int g2 = 0; g2 < G; ++g2) { a += part[(g2 * nch + threadIdx.x) * 2]; b += part[(g2 * nch + threadIdx.x) * 2 + 1]; }
            const int cq = threadIdx.x / V, e = threadIdx.x - cq * V;
            const int chunk = blockIdx.y * 256 + cq;
            if (chunk < cpp) {
                atomicAdd(dst + chunk * V + e, a);
                atomicAdd(dst + Cp + chunk * V + e, b);
            }
        }
        return;
    }
    for (j = threadIdx.x; j < nch; j += 256) {
        const int cq = j / V, e = j - cq * V;
        const int chunk = blockIdx.y * 256 + cq;
        if (chunk >= cpp) continue;
        a = 0.f; b = 0.f;
        for (int l = 0; l < L.R; ++l) {
            const int tt = l * L.cpb + cq;
            a += red[(tt * 2 + 0) * V + e];
            b += red[(tt * 2 + 1) * V + e];
        }
        atomicAdd(dst + chunk * V + e, a);
        atomicAdd(dst + Cp + chunk * V + e, b);
    }
}

template <typename T, int ACT, int RESM>
__global__ __launch_bounds__(256) void bn_bwd_apply_sums_kernel(const T* __restrict__ y, int ldy, const T* __restrict__ dout, int lddo,
                                                                const T* __restrict__ out, int ldo,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ sums, T* __restrict__ dy, int lddy,
                                                                T* __restrict__ dres, int lddr, int dres_acc, float* dgamma, float* dbeta,
                                                                int acc_param, long long npix, int C, int Cp) {
    constexpr int V = ET<T>::V;
    const Lay L = make_lay<V>(Cp);
    const float invM = 1.0f / (float)npix;
    __shared__ float s_kb[256 * 8], s_kg[256 * 8];
    {
        const int nch = L.cpb * V;
        const int cbase = blockIdx.y * 256 * V;
        for (int j = threadIdx.x; j < nch; j += 256) {
            const int c = cbase + j;
            float db = 0.f, dg = 0.f;
            if (c < Cp) {
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int r = 0; r < YDL_BN_REPLICAS; ++r) {
                    s1 += (double)sums[(size_t)(2 * r) * Cp + c];
                    s2 += (double)sums[(size_t)(2 * r + 1) * Cp + c];
                }
                db = (float)s1; dg = (float)s2;
                if (blockIdx.x == 0 && c < C) {
                    if (dbeta) dbeta[c] = (acc_param ? dbeta[c] : 0.f) + db;
                    if (dgamma) dgamma[c] = (acc_param ? dgamma[c] : 0.f) + dg;
                }
            }
            s_kb[j] = db * invM; s_kg[j] = dg * invM;
        }
    }
    __syncthreads();
    if (!L.live) return;
    float sc[V], sf[V], mu[V], is[V], kb[V], kg[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const int c = L.c + e;
        sc[e] = scale[c]; sf[e] = shift[c]; mu[e] = mean[c]; is[e] = invstd[c];
        kb[e] = s_kb[L.cq * V + e]; kg[e] = s_kg[L.cq * V + e];
    }
    const long long stride = (long long)gridDim.x * L.R;
    // the second pass walks the tensor BACKWARDS: what the reduce pass read last is the likeliest to still sit in L2 / the
    // Infinity Cache (BN_APPLY_REVERSE=0: same order as the reduce pass, for A/B timing)
    const long long pfirst = (long long)blockIdx.x * L.R + L.pl;
    const long long plast = pfirst < npix ? pfirst + (npix - 1 - pfirst) / stride * stride : pfirst - stride;
    for (long long pix = BN_APPLY_REVERSE ? plast : pfirst; BN_APPLY_REVERSE ? pix >= pfirst : pix < npix;
         pix += BN_APPLY_REVERSE ? -stride : stride) {
        float dz[V], xh[V], o[V], dv[V];
        dz_xhat<T, ACT>(y, dout, out, pix, ldy, lddo, ldo, L.c, sc, sf, mu, is, dz, xh, dv);
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = sc[e] * (dz[e] - kb[e] - xh[e] * kg[e]);
        *(uint4*)(dy + pix * lddy + L.c) = pack16<T>(o);
        if (RESM != 0) {
            T* rp = dres + pix * lddr + L.c;
            float r[V];
#pragma unroll
            for (int e = 0; e < V; ++e) r[e] = RESM == 1 ? dv[e] : dz[e];
            if (dres_acc) {
                float old[V];
                unpack16<T>(*(const uint4*)rp, old);
#pragma unroll
                for (int e = 0; e < V; ++e) r[e] += old[e];
            }
            *(uint4*)rp = pack16<T>(r);
        }
    }
}

static int bn_act_bwd_sums_impl(int dtype, const void* y, int ldy, const void* dout, int lddo, const void* out, int ldo,
                                const float* mean, const float* invstd, const float* scale, const float* shift,
                                int res_mode, int act, void* dy, int lddy, void* dres, int lddr,
                                float* dgamma, float* dbeta, int accumulate_param_grads,
                                float* sums, int64_t npix, int C, int Cp, void* stream, bool with_reduce) {
    const int dres_acc = (res_mode & YDL_RES_GRAD_ACCUMULATE) ? 1 : 0;
    const int rmode = res_mode & 15;
    YDL_CHECK(dtype == YDL_F32 || dtype == YDL_BF16, "bad dtype");
    YDL_CHECK(rmode == YDL_RES_NONE || rmode == YDL_RES_AFTER_ACT || rmode == YDL_RES_BEFORE_ACT, "unknown residual mode");
    const int resm = dres == nullptr ? 0 : (rmode == YDL_RES_AFTER_ACT ? 1 : 2);
    YDL_CHECK(dres == nullptr || (lddr >= Cp && aligned16(dres)), "dres must be 16-byte aligned with a stride covering Cp");
    const int V = dtype == YDL_F32 ? 4 : 8;
    YDL_CHECK(y && dout && dy && mean && invstd && scale && shift && sums, "null pointer");
    YDL_CHECK(act != YDL_ACT_RELU || out != nullptr, "RELU backward needs the saved output");
    YDL_CHECK(Cp > 0 && Cp % V == 0 && C <= Cp && ldy >= Cp && lddo >= Cp && lddy >= Cp, "bad channel geometry");
    YDL_CHECK(aligned16(y) && aligned16(dout) && aligned16(dy) && aligned16(sums), "16-byte alignment");
    YDL_CHECK(act == YDL_ACT_NONE || act == YDL_ACT_SILU || act == YDL_ACT_RELU, "unknown activation");
    hipStream_t st = (hipStream_t)stream;
    dim3 g1 = lay_grid(npix, Cp, V, BWD_MAX_PARTIALS);
    dim3 g3 = lay_grid(npix, Cp, V, 256 * 8);
    {
        // small tensors (40^2 / 20^2 layers): no more CTAs than pixel-row passes of eight (reduce: four 2-pixel iterations and an
        // eighth of the per-CTA atomic passes) / four (apply) — every CTA pays a coefficient prologue of about 2 us
        // (tools/bn_bench.py: 6400 x 512 19.8 -> 14.8 us, 6400 x 1024 26.5 -> 22.0 us; no change from 25 600 pixels up)
        static const int small = getenv("YDL_BN_SMALLGRID") ? atoi(getenv("YDL_BN_SMALLGRID")) : 1;
        static const int f1 = getenv("YDL_BN_SG_REDUCE") ? atoi(getenv("YDL_BN_SG_REDUCE")) : 8;
        static const int f3 = getenv("YDL_BN_SG_APPLY") ? atoi(getenv("YDL_BN_SG_APPLY")) : 4;
        const int cpp = Cp / V, cpb = cpp < 256 ? cpp : 256, R = 256 / cpb;
        const long long w1 = (npix + (long long)R * f1 - 1) / ((long long)R * f1), w3 = (npix + (long long)R * f3 - 1) / ((long long)R * f3);
        if (small && w1 < (long long)g1.x) g1.x = (unsigned)(w1 < 1 ? 1 : w1);
        if (small && w3 < (long long)g3.x) g3.x = (unsigned)(w3 < 1 ? 1 : w3);
    }
#define YDL_BS_APPLY(T, A, RM)                                                                                                     \
    bn_bwd_apply_sums_kernel<T, A, RM><<<g3, 256, 0, st>>>((const T*)y, ldy, (const T*)dout, lddo, (const T*)out, ldo, scale, shift, \
                                                           mean, invstd, sums, (T*)dy, lddy, (T*)dres, lddr, dres_acc, dgamma, dbeta, \
                                                           accumulate_param_grads, npix, C, Cp)
#define YDL_BS_LAUNCH(T, A)                                                                                                        \
    do {                                                                                                                           \
        if (with_reduce)                                                                                                           \
            bn_bwd_reduce_sums_kernel<T, A><<<g1, 256, 0, st>>>((const T*)y, ldy, (const T*)dout, lddo, (const T*)out, ldo, scale, \
                                                                shift, mean, invstd, sums, npix, Cp);                              \
        if (resm == 0) YDL_BS_APPLY(T, A, 0);                                                                                      \
        else if (resm == 1) YDL_BS_APPLY(T, A, 1);                                                                                 \
        else YDL_BS_APPLY(T, A, 2);                                                                                                \
    } while (0)
#define YDL_BS_ACT(T)                                                   \
    do {                                                                \
        if (act == YDL_ACT_SILU) YDL_BS_LAUNCH(T, YDL_ACT_SILU);        \
        else if (act == YDL_ACT_RELU) YDL_BS_LAUNCH(T, YDL_ACT_RELU);   \
        else YDL_BS_LAUNCH(T, YDL_ACT_NONE);                            \
    } while (0)
    if (dtype == YDL_F32) YDL_BS_ACT(float);
    else YDL_BS_ACT(bf16_t);
    YDL_LAUNCH_CHECK();
    return 0;
}

extern "C" int ydl_bn_act_bwd_sums(int dtype, const void* y, int ldy, const void* dout, int lddo, const void* out, int ldo,
                                   const float* mean, const float* invstd, const float* scale, const float* shift,
                                   int res_mode, int act, void* dy, int lddy, void* dres, int lddr,
                                   float* dgamma, float* dbeta, int accumulate_param_grads,
                                   float* sums, int64_t npix, int C, int Cp, void* stream) {
    return bn_act_bwd_sums_impl(dtype, y, ldy, dout, lddo, out, ldo, mean, invstd, scale, shift, res_mode, act, dy, lddy, dres, lddr,
                                dgamma, dbeta, accumulate_param_grads, sums, npix, C, Cp, stream, true);
}
extern "C" int ydl_bn_act_bwd_apply_sums(int dtype, const void* y, int ldy, const void* dout, int lddo, const void* out, int ldo,
                                         const float* mean, const float* invstd, const float* scale, const float* shift,
                                         int res_mode, int act, void* dy, int lddy, void* dres, int lddr,
                                         float* dgamma, float* dbeta, int accumulate_param_grads,
                                         float* sums, int64_t npix, int C, int Cp, void* stream) {
    return bn_act_bwd_sums_impl(dtype, y, ldy, dout, lddo, out, ldo, mean, invstd, scale, shift, res_mode, act, dy, lddy, dres, lddr,
                                dgamma, dbeta, accumulate_param_grads, sums, npix, C, Cp, stream, false);
}
