// Spatial NHWC ops: max-pool, nearest / bilinear resize (both align_corners conventions), strided slice copy,
// NCHW<->NHWC edge conversion, per-(n,c) channel gating.  HBM-bound, 16-byte channel vectors per thread.
// Index arithmetic restates ATen's (UpSample.h area_pixel_compute_source_index / nearest_idx) with explicit
// round-to-nearest f32 ops so the compiler cannot contract them into FMAs: indices are bit-exact vs the CPU.
#include "common.h"
#include <stdlib.h>

static inline int sgrid(long long total) {
    long long b = (total + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

// ------------------------------------------------------------------------------------------------------
// max pool
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                          uint8_t* __restrict__ idx, int N, int Hi, int Wi, int Ho, int Wo,
                                                          int Cp, int k, int s, int p) {
    constexpr int V = ET<T>::V;
    const int cpp = Cp / V;
    const long long total = (long long)N * Ho * Wo * cpp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int cq = (int)(i % cpp);
        long long pix = i / cpp;
        int wo = (int)(pix % Wo);
        long long t2 = pix / Wo;
        int ho = (int)(t2 % Ho);
        int n = (int)(t2 / Ho);
        float best[V];
        int bi[V];
#pragma unroll
        for (int e = 0; e < V; ++e) { best[e] = -INFINITY; bi[e] = 0; }
        bool first = true;
        for (int ky = 0; ky < k; ++ky) {
            int ih = ho * s - p + ky;
            if ((unsigned)ih >= (unsigned)Hi) continue;
            for (int kx = 0; kx < k; ++kx) {
                int iw = wo * s - p + kx;
                if ((unsigned)iw >= (unsigned)Wi) continue;
                float v[V];
                unpack16<T>(*(const uint4*)(x + ((size_t)(n * Hi + ih) * Wi + iw) * ldx + cq * V), v);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    // ATen: update when (val > max) || isnan(val); first valid element initialises
                    if (first || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = ky * k + kx; }
                }
                first = false;
            }
        }
        *(uint4*)(y + (size_t)pix * ldy + cq * V) = pack16<T>(best);
        if (idx) {
#pragma unroll
            for (int e = 0; e < V; ++e) idx[(size_t)pix * Cp + cq * V + e] = (uint8_t)bi[e];
        }
    }
}

// gather formulation: every input element collects dy from the (<= ceil(k/s)^2) windows whose arg-max it is
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, int lddy, const uint8_t* __restrict__ idx,
                                                          T* __restrict__ dx, int lddx, int accumulate, int N, int Hi, int Wi,
                                                          int Ho, int Wo, int Cp, int k, int s, int p) {
    constexpr int V = ET<T>::V;
    const int cpp = Cp / V;
    const long long total = (long long)N * Hi * Wi * cpp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int cq = (int)(i % cpp);
        long long pix = i / cpp;
        int iw = (int)(pix % Wi);
        long long t2 = pix / Wi;
        int ih = (int)(t2 % Hi);
        int n = (int)(t2 / Hi);
        float g[V];
#pragma unroll
        for (int e = 0; e < V; ++e) g[e] = 0.f;
        if (accumulate) unpack16<T>(*(const uint4*)(dx + (size_t)pix * lddx + cq * V), g);
        for (int ky = 0; ky < k; ++ky) {
            int nh = ih + p - ky;
            if (nh < 0 || nh % s != 0) continue;
            int ho = nh / s;
            if (ho >= Ho) continue;
            for (int kx = 0; kx < k; ++kx) {
                int nw = iw + p - kx;
                if (nw < 0 || nw % s != 0) continue;
                int wo = nw / s;
                if (wo >= Wo) continue;
                size_t op = ((size_t)(n * Ho + ho) * Wo + wo);
                float d[V];
                unpack16<T>(*(const uint4*)(dy + op * lddy + cq * V), d);
                const uint8_t* ip = idx + op * Cp + cq * V;
                int code = ky * k + kx;
#pragma unroll
                for (int e = 0; e < V; ++e) g[e] += (ip[e] == code) ? d[e] : 0.f;
            }
        }
        *(uint4*)(dx + (size_t)pix * lddx + cq * V) = pack16<T>(g);
    }
}

extern "C" int ydl_maxpool_fwd(int dtype, const void* x, int ldx, void* y, int ldy, uint8_t* idx,
                               int N, int Hi, int Wi, int Ho, int Wo, int C, int k, int s, int p, void* stream) {
    const int V = dtype == YDL_F32 ? 4 : 8;
    const int Cp = round_up(C, V);
    YDL_CHECK(x && y && ldx >= Cp && ldy >= Cp && k * k <= 255, "bad arguments");
    YDL_CHECK(Ho == (Hi + 2 * p - k) / s + 1 && Wo == (Wi + 2 * p - k) / s + 1, "output size mismatch");
    hipStream_t st = (hipStream_t)stream;
    int grid = sgrid((long long)N * Ho * Wo * (Cp / V));
    if (dtype == YDL_F32) maxpool_fwd_kernel<float><<<grid, 256, 0, st>>>((const float*)x, ldx, (float*)y, ldy, idx, N, Hi, Wi, Ho, Wo, Cp, k, s, p);
    else maxpool_fwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, ldx, (bf16_t*)y, ldy, idx, N, Hi, Wi, Ho, Wo, Cp, k, s, p);
    YDL_LAUNCH_CHECK();
    return 0;
}
extern "C" int ydl_maxpool_bwd(int dtype, const void* dy, int lddy, const uint8_t* idx, void* dx, int lddx, int accumulate,
                               int N, int Hi, int Wi, int Ho, int Wo, int C, int k, int s, int p, void* stream) {
    const int V = dtype == YDL_F32 ? 4 : 8;
    const int Cp = round_up(C, V);
    YDL_CHECK(dy && dx && idx && lddy >= Cp && lddx >= Cp, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int grid = sgrid((long long)N * Hi * Wi * (Cp / V));
    if (dtype == YDL_F32) maxpool_bwd_kernel<float><<<grid, 256, 0, st>>>((const float*)dy, lddy, idx, (float*)dx, lddx, accumulate, N, Hi, Wi, Ho, Wo, Cp, k, s, p);
    else maxpool_bwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)dy, lddy, idx, (bf16_t*)dx, lddx, accumulate, N, Hi, Wi, Ho, Wo, Cp, k, s, p);
    YDL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// SPPF's three chained k x k / stride 1 max-pools (seg_diceloss_yolov5.py:468-481: y1 = mp(x), y2 = mp(y1), y3 = mp(y2)) in ONE
// launch per direction.  A CTA owns the whole H x W plane of one image for SP_CG 16-byte channel chunks and keeps it in LDS, so
// the chain never goes back to memory between the pools (three launches of 22-25 us forward and 30-34 us backward on 6.5 MB
// tensors: latency, not bytes).  Values, arg-max codes and the backward's summation order are those of maxpool_fwd_kernel /
// maxpool_bwd_kernel (first maximum in scan order; every stage rounded to the storage type) — the results are bit-identical.
// ------------------------------------------------------------------------------------------------------
// V arg-max codes (one byte each) of an item as one LDS word / double word
template <int V> struct SpCodes;
template <> struct SpCodes<8> {
    typedef uint2 W;
    __device__ static __forceinline__ W pack(const int* c) {
        return make_uint2((unsigned)c[0] | ((unsigned)c[1] << 8) | ((unsigned)c[2] << 16) | ((unsigned)c[3] << 24),
                          (unsigned)c[4] | ((unsigned)c[5] << 8) | ((unsigned)c[6] << 16) | ((unsigned)c[7] << 24));
    }
    __device__ static __forceinline__ int get(const W& w, int e) { return (int)(((e < 4 ? w.x : w.y) >> ((e & 3) * 8)) & 0xffu); }
};
template <> struct SpCodes<4> {
    typedef unsigned W;
    __device__ static __forceinline__ W pack(const int* c) {
        return (unsigned)c[0] | ((unsigned)c[1] << 8) | ((unsigned)c[2] << 16) | ((unsigned)c[3] << 24);
    }
    __device__ static __forceinline__ int get(const W& w, int e) { return (int)((w >> (e * 8)) & 0xffu); }
};
#define SP_CG 2
// Forward: every pool is SEPARABLE — a row pass keeps (row maximum, kx of its first occurrence) per position, a column pass takes the
// first row whose maximum beats the running one: k + k window taps instead of k * k (the pools are VALU-bound on the compare /
// select chain: 62 us of lane operations for the three 5x5 pools of BASELINE config 2), and the result is still the FIRST maximum in
// (ky, kx) scan order with ATen's update rule `v > best || isnan(v)` applied along both passes (a NaN wins and the last one stays,
// in either formulation; the first in-range tap initialises).
template <typename T>
__global__ __launch_bounds__(256) void sppf_pool_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y1, T* __restrict__ y2,
                                                            T* __restrict__ y3, int ldy, uint8_t* __restrict__ i1, uint8_t* __restrict__ i2,
                                                            uint8_t* __restrict__ i3, int H, int W, int Cp, int k) {
    constexpr int V = ET<T>::V;
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    const int HW = H * W;
    const int items = HW * SP_CG;
    uint4* cur = (uint4*)sp_smem;                         // the pool's input, then its output
    uint4* rmax = cur + items;                            // row maxima
    typedef SpCodes<V> CW;
    typename CW::W* rkx = (typename CW::W*)(rmax + items);          // [items] kx of the row maxima, one byte per channel
    const int cpp = Cp / V;
    const int n = blockIdx.y;
    const int cq0 = blockIdx.x * SP_CG;
    const int p = k / 2;
    for (int it = threadIdx.x; it < items; it += 256) {
        const int cg = it % SP_CG, pix = it / SP_CG;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (cq0 + cg < cpp) v = *(const uint4*)(x + ((size_t)n * HW + pix) * ldx + (cq0 + cg) * V);
        cur[it] = v;
    }
    __syncthreads();
    T* const ys[3] = {y1, y2, y3};
    uint8_t* const is[3] = {i1, i2, i3};
#pragma unroll
    for (int st = 0; st < 3; ++st) {
        for (int it = threadIdx.x; it < items; it += 256) {               // row pass
            const int cg = it % SP_CG, pix = it / SP_CG;
            const int wo = pix % W, ho = pix / W;
            float best[V];
            int bk[V];
#pragma unroll
            for (int e = 0; e < V; ++e) { best[e] = -INFINITY; bk[e] = 0; }
            bool first = true;
            for (int kx = 0; kx < k; ++kx) {
                const int iw = wo - p + kx;
                if ((unsigned)iw >= (unsigned)W) continue;
                float v[V];
                unpack16<T>(cur[(ho * W + iw) * SP_CG + cg], v);
#pragma unroll
                for (int e = 0; e < V; ++e)
                    if (first || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bk[e] = kx; }
                first = false;
            }
            rmax[it] = pack16<T>(best);
            rkx[it] = CW::pack(bk);
        }
        __syncthreads();
        for (int it = threadIdx.x; it < items; it += 256) {               // column pass
            const int cg = it % SP_CG, pix = it / SP_CG;
            const int wo = pix % W, ho = pix / W;
            float best[V];
            int bi[V];
#pragma unroll
            for (int e = 0; e < V; ++e) { best[e] = -INFINITY; bi[e] = 0; }
            bool first = true;
            for (int ky = 0; ky < k; ++ky) {
                const int ih = ho - p + ky;
                if ((unsigned)ih >= (unsigned)H) continue;
                const int rit = (ih * W + wo) * SP_CG + cg;
                float v[V];
                unpack16<T>(rmax[rit], v);
                const typename CW::W rk = rkx[rit];
#pragma unroll
                for (int e = 0; e < V; ++e)
                    if (first || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = ky * k + CW::get(rk, e); }
                first = false;
            }
            const uint4 o = pack16<T>(best);
            if (st < 2) cur[it] = o;
            if (cq0 + cg < cpp) {
                const size_t gp = (size_t)n * HW + pix;
                *(uint4*)(ys[st] + gp * ldy + (cq0 + cg) * V) = o;
                if (is[st]) *(typename CW::W*)(is[st] + gp * Cp + (cq0 + cg) * V) = CW::pack(bi);
            }
        }
        __syncthreads();
    }
}

// backward of the chain: g2 = dy2 + mp'(dy3), g1 = dy1 + mp'(g2), dx (+)= mp'(g1); every stage is the gather of maxpool_bwd_kernel
template <typename T>
__global__ __launch_bounds__(256) void sppf_pool_bwd_kernel(const T* __restrict__ dy1, const T* __restrict__ dy2, const T* __restrict__ dy3,
                                                            int lddy, const uint8_t* __restrict__ i1, const uint8_t* __restrict__ i2,
                                                            const uint8_t* __restrict__ i3, T* __restrict__ dx, int lddx, int accumulate,
                                                            int H, int W, int Cp, int k) {
    constexpr int V = ET<T>::V;
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    const int HW = H * W;
    uint4* plane[2] = {(uint4*)sp_smem, (uint4*)sp_smem + (size_t)HW * SP_CG};
    typedef SpCodes<V> CW;
    typename CW::W* icode = (typename CW::W*)((uint4*)sp_smem + (size_t)2 * HW * SP_CG);   // [items] arg-max codes of the stage
    const int cpp = Cp / V;
    const int n = blockIdx.y;
    const int cq0 = blockIdx.x * SP_CG;
    const int p = k / 2;
    const int items = HW * SP_CG;
    const T* const dys[3] = {dy3, dy2, dy1};          // gradient that enters stage st from its own output slice
    const uint8_t* const is[3] = {i3, i2, i1};
    for (int it = threadIdx.x; it < items; it += 256) {
        const int cg = it % SP_CG, pix = it / SP_CG;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (cq0 + cg < cpp) v = *(const uint4*)(dy3 + ((size_t)n * HW + pix) * lddy + (cq0 + cg) * V);
        plane[0][it] = v;
    }
#pragma unroll
    for (int st = 0; st < 3; ++st) {
        // codes of this stage's pool (output positions)
        for (int it = threadIdx.x; it < items; it += 256) {
            const int cg = it % SP_CG, pix = it / SP_CG;
            if (cq0 + cg < cpp) icode[it] = *(const typename CW::W*)(is[st] + ((size_t)n * HW + pix) * Cp + (cq0 + cg) * V);
        }
        __syncthreads();
        const uint4* src = plane[st & 1];
        uint4* dst = plane[(st + 1) & 1];
        for (int it = threadIdx.x; it < items; it += 256) {
            const int cg = it % SP_CG, pix = it / SP_CG;
            if (cq0 + cg >= cpp) continue;
            const int iw = pix % W, ih = pix / W;
            const size_t gp = (size_t)n * HW + pix;
            float g[V];
#pragma unroll
            for (int e = 0; e < V; ++e) g[e] = 0.f;
            // what the unfused chain finds in the buffer it accumulates into: the next slice's own gradient (always there), or
            // the previous contents of dx for the last stage
            if (st < 2) unpack16<T>(*(const uint4*)(dys[st + 1] + gp * lddy + (cq0 + cg) * V), g);
            else if (accumulate) unpack16<T>(*(const uint4*)(dx + gp * lddx + (cq0 + cg) * V), g);
            for (int ky = 0; ky < k; ++ky) {
                const int ho = ih + p - ky;
                if (ho < 0 || ho >= H) continue;
                for (int kx = 0; kx < k; ++kx) {
                    const int wo = iw + p - kx;
                    if (wo < 0 || wo >= W) continue;
                    const int oit = (ho * W + wo) * SP_CG + cg;
                    float d[V];
                    unpack16<T>(src[oit], d);
                    const typename CW::W cw = icode[oit];
                    const int code = ky * k + kx;
#pragma unroll
                    for (int e = 0; e < V; ++e) g[e] += (CW::get(cw, e) == code) ? d[e] : 0.f;
                }
            }
            const uint4 o = pack16<T>(g);
            if (st < 2) dst[it] = o;
            else *(uint4*)(dx + gp * lddx + (cq0 + cg) * V) = o;
        }
        __syncthreads();
    }
}

static inline size_t sppf_smem(int dtype, int H, int W, bool bwd) {
    const int V = dtype == YDL_F32 ? 4 : 8;
    (void)bwd;
    return (size_t)H * W * SP_CG * (2 * 16 + V);          // forward: input/output plane, row maxima, their kx; backward: two planes, codes
}
extern "C" int ydl_sppf_pool_supported(int dtype, int H, int W, int C, int k) {
    if (dtype != YDL_F32 && dtype != YDL_BF16) return 0;
    if (k < 1 || k % 2 == 0 || k * k > 255 || H < 1 || W < 1 || C < 1) return 0;
    return sppf_smem(dtype, H, W, true) <= 64 * 1024 ? 1 : 0;
}
extern "C" int ydl_sppf_pool_fwd(int dtype, const void* x, int ldx, void* y1, void* y2, void* y3, int ldy,
                                 uint8_t* idx1, uint8_t* idx2, uint8_t* idx3, int N, int H, int W, int C, int k, void* stream) {
    const int V = dtype == YDL_F32 ? 4 : 8;
    const int Cp = round_up(C, V);
    YDL_CHECK(ydl_sppf_pool_supported(dtype, H, W, C, k), "plane too large for the LDS-resident form (query ydl_sppf_pool_supported)");
    YDL_CHECK(x && y1 && y2 && y3 && ldx >= Cp && ldy >= Cp && N >= 1, "bad arguments");
    YDL_CHECK((idx1 == nullptr) == (idx2 == nullptr) && (idx2 == nullptr) == (idx3 == nullptr), "the three index planes come together");
    YDL_CHECK(aligned16(x) && aligned16(y1) && aligned16(y2) && aligned16(y3) && ldx % V == 0 && ldy % V == 0, "16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((Cp / V + SP_CG - 1) / SP_CG, N);
    const size_t smem = sppf_smem(dtype, H, W, false);
    if (dtype == YDL_F32) sppf_pool_fwd_kernel<float><<<grid, 256, smem, st>>>((const float*)x, ldx, (float*)y1, (float*)y2, (float*)y3, ldy, idx1, idx2, idx3, H, W, Cp, k);
    else sppf_pool_fwd_kernel<bf16_t><<<grid, 256, smem, st>>>((const bf16_t*)x, ldx, (bf16_t*)y1, (bf16_t*)y2, (bf16_t*)y3, ldy, idx1, idx2, idx3, H, W, Cp, k);
    YDL_LAUNCH_CHECK();
    return 0;
}
extern "C" int ydl_sppf_pool_bwd(int dtype, const void* dy1, const void* dy2, const void* dy3, int lddy, const uint8_t* idx1,
                                 const uint8_t* idx2, const uint8_t* idx3, void* dx, int lddx, int accumulate,
                                 int N, int H, int W, int C, int k, void* stream) {
    const int V = dtype == YDL_F32 ? 4 : 8;
    const int Cp = round_up(C, V);
    YDL_CHECK(ydl_sppf_pool_supported(dtype, H, W, C, k), "plane too large for the LDS-resident form (query ydl_sppf_pool_supported)");
    YDL_CHECK(dy1 && dy2 && dy3 && idx1 && idx2 && idx3 && dx && lddy >= Cp && lddx >= Cp && N >= 1, "bad arguments");
    YDL_CHECK(aligned16(dy1) && aligned16(dy2) && aligned16(dy3) && aligned16(dx) && lddy % V == 0 && lddx % V == 0, "16-byte alignment");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((Cp / V + SP_CG - 1) / SP_CG, N);
    const size_t smem = sppf_smem(dtype, H, W, true);
    if (dtype == YDL_F32) sppf_pool_bwd_kernel<float><<<grid, 256, smem, st>>>((const float*)dy1, (const float*)dy2, (const float*)dy3, lddy, idx1, idx2, idx3, (float*)dx, lddx, accumulate, H, W, Cp, k);
    else sppf_pool_bwd_kernel<bf16_t><<<grid, 256, smem, st>>>((const bf16_t*)dy1, (const bf16_t*)dy2, (const bf16_t*)dy3, lddy, idx1, idx2, idx3, (bf16_t*)dx, lddx, accumulate, H, W, Cp, k);
    YDL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// resize
// ------------------------------------------------------------------------------------------------------
struct Lin { int i0, i1; float w0, w1; };

// bilinear source index/weights for one axis, ATen semantics (float accscalar)
__device__ __forceinline__ Lin lin_src(int dst, float scale, int in, bool align) {
    Lin L;
    float src;
    if (align) {
        src = __fmul_rn(scale, (float)dst);
    } else {
        src = __fsub_rn(__fmul_rn(scale, __fadd_rn((float)dst, 0.5f)), 0.5f);
        if (src < 0.f) src = 0.f;
    }
    L.i0 = (int)src;
    if (L.i0 > in - 1) L.i0 = in - 1;
    L.i1 = L.i0 + (L.i0 < in - 1 ? 1 : 0);
    L.w1 = __fsub_rn(src, (float)L.i0);
    L.w0 = __fsub_rn(1.f, L.w1);
    return L;
}
__device__ __forceinline__ int near_src(int dst, float scale, int in) {
    int v = (int)floorf(__fmul_rn((float)dst, scale));
    return v < in - 1 ? v : in - 1;
}

template <typename T>
__global__ __launch_bounds__(256) void resize_fwd_kernel(int mode, const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                         int N, int Hi, int Wi, int Ho, int Wo, int Cp, float sh, float sw) {
    constexpr int V = ET<T>::V;
    const int cpp = Cp / V;
    const long long total = (long long)N * Ho * Wo * cpp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int cq = (int)(i % cpp);
        long long pix = i / cpp;
        int wo = (int)(pix % Wo);
        long long t2 = pix / Wo;
        int ho = (int)(t2 % Ho);
        int n = (int)(t2 / Ho);
        const T* xb = x + (size_t)n * Hi * Wi * ldx + cq * V;
        float o[V];
        if (mode == 0) {
            int ih = near_src(ho, sh, Hi), iw = near_src(wo, sw, Wi);
            *(uint4*)(y + (size_t)pix * ldy + cq * V) = *(const uint4*)(xb + ((size_t)ih * Wi + iw) * ldx);
            continue;
        }
        Lin lh = lin_src(ho, sh, Hi, mode == 2), lw = lin_src(wo, sw, Wi, mode == 2);
        float v00[V], v01[V], v10[V], v11[V];
        unpack16<T>(*(const uint4*)(xb + ((size_t)lh.i0 * Wi + lw.i0) * ldx), v00);
        unpack16<T>(*(const uint4*)(xb + ((size_t)lh.i0 * Wi + lw.i1) * ldx), v01);
        unpack16<T>(*(const uint4*)(xb + ((size_t)lh.i1 * Wi + lw.i0) * ldx), v10);
        unpack16<T>(*(const uint4*)(xb + ((size_t)lh.i1 * Wi + lw.i1) * ldx), v11);
#pragma unroll
        for (int e = 0; e < V; ++e)
            o[e] = lh.w0 * (lw.w0 * v00[e] + lw.w1 * v01[e]) + lh.w1 * (lw.w0 * v10[e] + lw.w1 * v11[e]);
        *(uint4*)(y + (size_t)pix * ldy + cq * V) = pack16<T>(o);
    }
}

// gather backward: an input element (ih, iw) scans the output range that can reference it and re-derives the
// forward index/weights for each candidate => exactly the transpose of the forward, for any scale.
__device__ __forceinline__ void cand_range(int i, int in, int out, float scale, int mode, int& lo, int& hi) {
    // conservative bounds on {dst : src(dst) in (i-1, i+1)}: dst ~ (i +- 1 + 0.5)/scale
    float inv = scale > 0.f ? 1.0f / scale : (float)out;
    float a = ((float)i - 1.5f) * inv - 2.f, b = ((float)i + 1.5f) * inv + 2.f;
    if (mode == 0) { a = (float)i * inv - 2.f; b = ((float)i + 1.f) * inv + 2.f; }
    lo = a < 0.f ? 0 : (int)a;
    hi = b > (float)(out - 1) ? out - 1 : (int)b;
    if (i == in - 1) hi = out - 1;      // clamped tail (nearest min(), bilinear edge clamp)
    if (i == 0) lo = 0;
}

template <typename T>
__global__ __launch_bounds__(256) void resize_bwd_kernel(int mode, const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx,
                                                         int accumulate, int N, int Hi, int Wi, int Ho, int Wo, int Cp, float sh, float sw) {
    constexpr int V = ET<T>::V;
    const int cpp = Cp / V;
    const long long total = (long long)N * Hi * Wi * cpp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int cq = (int)(i % cpp);
        long long pix = i / cpp;
        int iw = (int)(pix % Wi);
        long long t2 = pix / Wi;
        int ih = (int)(t2 % Hi);
        int n = (int)(t2 / Hi);
        float g[V];
#pragma unroll
        for (int e = 0; e < V; ++e) g[e] = 0.f;
        if (accumulate) unpack16<T>(*(const uint4*)(dx + (size_t)pix * lddx + cq * V), g);
        int hlo, hhi, wlo, whi;
        cand_range(ih, Hi, Ho, sh, mode, hlo, hhi);
        cand_range(iw, Wi, Wo, sw, mode, wlo, whi);
        const T* db = dy + (size_t)n * Ho * Wo * lddy + cq * V;
        for (int ho = hlo; ho <= hhi; ++ho) {
            float wh;
            if (mode == 0) wh = near_src(ho, sh, Hi) == ih ? 1.f : 0.f;
            else { Lin l = lin_src(ho, sh, Hi, mode == 2); wh = (l.i0 == ih ? l.w0 : 0.f) + (l.i1 == ih ? l.w1 : 0.f); }
            if (wh == 0.f) continue;
            for (int wo = wlo; wo <= whi; ++wo) {
                float ww;
                if (mode == 0) ww = near_src(wo, sw, Wi) == iw ? 1.f : 0.f;
                else { Lin l = lin_src(wo, sw, Wi, mode == 2); ww = (l.i0 == iw ? l.w0 : 0.f) + (l.i1 == iw ? l.w1 : 0.f); }
                if (ww == 0.f) continue;
                float d[V];
                unpack16<T>(*(const uint4*)(db + ((size_t)ho * Wo + wo) * lddy), d);
                float wgt = wh * ww;
#pragma unroll
                for (int e = 0; e < V; ++e) g[e] += wgt * d[e];
            }
        }
        *(uint4*)(dx + (size_t)pix * lddx + cq * V) = pack16<T>(g);
    }
}

// Bilinear (align_corners = False) up-sampling by an INTEGER factor S: the outputs that reference input index i are exactly
// S*i - S/2 ... S*i + S + S/2 - 1 (2S candidates per axis, clamped to the image), so the gather backward needs 2S + 2S index/weight
// derivations per element instead of one per candidate PAIR of the generic kernel above (a 16 x 16 candidate window at S = 4: the
// generic form is VALU-bound, 82 us for the 128-channel 160^2 -> 40^2 gradient of BASELINE config 2).  Index and weights still come
// from lin_src, i.e. the transpose of the forward kernel's arithmetic, border clamps included.
template <typename T, int S>
__global__ __launch_bounds__(256) void resize_bwd_int_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, int accumulate,
                                                             int N, int Hi, int Wi, int Cp, float sh, float sw) {
    constexpr int V = ET<T>::V;
    constexpr int NC = 2 * S;
    const int Ho = Hi * S, Wo = Wi * S;
    const int cpp = Cp / V;
    const long long total = (long long)N * Hi * Wi * cpp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cq = (int)(i % cpp);
        const long long pix = i / cpp;
        const int iw = (int)(pix % Wi);
        const long long t2 = pix / Wi;
        const int ih = (int)(t2 % Hi);
        const int n = (int)(t2 / Hi);
        float g[V];
#pragma unroll
        for (int e = 0; e < V; ++e) g[e] = 0.f;
        if (accumulate) unpack16<T>(*(const uint4*)(dx + (size_t)pix * lddx + cq * V), g);
        float wwv[NC];
#pragma unroll
        for (int b = 0; b < NC; ++b) {
            const int wo = S * iw - S / 2 + b;
            float ww = 0.f;
            if ((unsigned)wo < (unsigned)Wo) { const Lin l = lin_src(wo, sw, Wi, false); ww = (l.i0 == iw ? l.w0 : 0.f) + (l.i1 == iw ? l.w1 : 0.f); }
            wwv[b] = ww;
        }
        const T* db = dy + (size_t)n * Ho * Wo * lddy + cq * V;
#pragma unroll
        for (int a = 0; a < NC; ++a) {
            const int ho = S * ih - S / 2 + a;
            if ((unsigned)ho >= (unsigned)Ho) continue;
            const Lin l = lin_src(ho, sh, Hi, false);
            const float wh = (l.i0 == ih ? l.w0 : 0.f) + (l.i1 == ih ? l.w1 : 0.f);
            if (wh == 0.f) continue;
#pragma unroll
            for (int b = 0; b < NC; ++b) {
                const int wo = S * iw - S / 2 + b;
                if (wwv[b] == 0.f) continue;             // (also every column outside the image)
                float d[V];
                unpack16<T>(*(const uint4*)(db + ((size_t)ho * Wo + wo) * lddy), d);
                const float wgt = wh * wwv[b];
#pragma unroll
                for (int e = 0; e < V; ++e) g[e] += wgt * d[e];
            }
        }
        *(uint4*)(dx + (size_t)pix * lddx + cq * V) = pack16<T>(g);
    }
}

// y += bilinear_resize(x), and (optionally) the per-channel (sum, sum of squares) of the RESULT added to BatchNorm replica rows: the
// last launch of a 1x1 convolution over a virtual concat ``conv_a(a) + resize(conv_b(b))`` (tape.conv_bn_act), when it is the
// resize that comes last.  Before, the resize initialised y (a full write) and the convolution accumulated into it through the
// point-wise kernel's register-layout read-modify-write with statistics (94 us instead of 48 for the plain launch at 128 ch @ 160^2);
// here the convolution writes plainly and this streaming pass does the read-modify-write.  Statistics are taken from the f32 sums
// before they are rounded to the storage type, as the convolution epilogues do.  Thread layout as the BatchNorm streaming kernels:
// a thread owns one 16-byte channel chunk and walks pixels.
template <typename T, bool ACC>      // ACC: y += resize(x) with optional statistics; otherwise y = resize(x) (ydl_resize_fwd)
__global__ __launch_bounds__(256) void resize_acc_sums_kernel(int mode, const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy,
                                                              int Hi, int Wi, int Ho, int Wo, int Cp, float sh, float sw,
                                                              int nrows, float* __restrict__ sums, int sums_ld) {
    constexpr int V = ET<T>::V;
    const int cpp = Cp / V;
    const int cpb = cpp < 256 ? cpp : 256;
    const int R = 256 / cpb;
    const int cq = threadIdx.x % cpb, pl = threadIdx.x / cpb;
    const int chunk = blockIdx.y * 256 + cq;
    const bool live = pl < R && chunk < cpp;
    const int c0 = chunk * V;
    float s1[V], s2[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    // a CTA walks whole output rows (n, ho): the row's source rows and weights once per row, no 64-bit index arithmetic per element
    // (the first version decoded a 64-bit pixel index per element: 115 us instead of 45 for the 128-channel 160^2 case)
    if (live) {
        for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
            const int n = row / Ho, ho = row - n * Ho;
            const T* xb = x + (size_t)n * Hi * Wi * ldx + c0;
            T* yrow = y + (size_t)row * Wo * ldy + c0;
            Lin lh;
            int ihn = 0;
            if (mode == 0) ihn = near_src(ho, sh, Hi);
            else lh = lin_src(ho, sh, Hi, mode == 2);
            const T* r0 = xb + (size_t)(mode == 0 ? ihn : lh.i0) * Wi * ldx;
            const T* r1 = xb + (size_t)(mode == 0 ? ihn : lh.i1) * Wi * ldx;
            for (int wo = pl; wo < Wo; wo += R) {
                float o[V], yv[V];
                if (mode == 0) {
                    unpack16<T>(*(const uint4*)(r0 + (size_t)near_src(wo, sw, Wi) * ldx), o);
                } else {
                    const Lin lw = lin_src(wo, sw, Wi, mode == 2);
                    float v00[V], v01[V], v10[V], v11[V];
                    unpack16<T>(*(const uint4*)(r0 + (size_t)lw.i0 * ldx), v00);
                    unpack16<T>(*(const uint4*)(r0 + (size_t)lw.i1 * ldx), v01);
                    unpack16<T>(*(const uint4*)(r1 + (size_t)lw.i0 * ldx), v10);
                    unpack16<T>(*(const uint4*)(r1 + (size_t)lw.i1 * ldx), v11);
#pragma unroll
                    for (int e = 0; e < V; ++e)
                        o[e] = lh.w0 * (lw.w0 * v00[e] + lw.w1 * v01[e]) + lh.w1 * (lw.w0 * v10[e] + lw.w1 * v11[e]);
                }
                T* yp = yrow + (size_t)wo * ldy;
                if constexpr (ACC) {
                    unpack16<T>(*(const uint4*)yp, yv);
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        const float v = yv[e] + o[e];
                        yv[e] = v;
                        s1[e] += v;
                        s2[e] = fmaf(v, v, s2[e]);
                    }
                    *(uint4*)yp = pack16<T>(yv);
                } else {
                    *(uint4*)yp = pack16<T>(o);
                }
            }
        }
    }
    if (!ACC || sums == nullptr) return;
    __shared__ float red[256 * 2 * 8];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        red[(threadIdx.x * 2 + 0) * V + e] = live ? s1[e] : 0.f;
        red[(threadIdx.x * 2 + 1) * V + e] = live ? s2[e] : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < cpb && chunk < cpp) {
        float* dst = sums + (size_t)(blockIdx.x & (YDL_BN_REPLICAS - 1)) * 2 * sums_ld;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float a = 0.f, b = 0.f;
            for (int l = 0; l < R; ++l) {
                const int tt = l * cpb + threadIdx.x;
                a += red[(tt * 2 + 0) * V + e];
                b += red[(tt * 2 + 1) * V + e];
            }
            atomicAdd(dst + c0 + e, a);
            atomicAdd(dst + sums_ld + c0 + e, b);
        }
    }
}

int g_resize_rows = 1;         // ydl_debug_set key 11: 0 = the element-indexed forward kernel (tests)
int g_resize_int = 1;          // ydl_debug_set key 10: 0 = the generic gather backward also for integer scale factors (tests)
static inline float axis_scale(int mode, int in, int out, float given) {
    if (mode == 2) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
    if (given > 0.f) return given;
    return (float)in / (float)out;
}

extern "C" int ydl_resize_fwd(int dtype, int mode, const void* x, int ldx, void* y, int ldy,
                              int N, int Hi, int Wi, int Ho, int Wo, int C, float scale_h, float scale_w, void* stream) {
    const int V = dtype == YDL_F32 ? 4 : 8;
    const int Cp = round_up(C, V);
    YDL_CHECK(x && y && mode >= 0 && mode <= 2 && ldx >= Cp && ldy >= Cp, "bad arguments");
    float sh = axis_scale(mode, Hi, Ho, scale_h), sw = axis_scale(mode, Wi, Wo, scale_w);
    hipStream_t st = (hipStream_t)stream;
    if (g_resize_rows && (long long)N * Ho < (1ll << 30)) {
        // row-walking form (resize_acc_sums_kernel<T, false>): same arithmetic per element, no 64-bit index decode per element
        const int nrows = N * Ho, cpp = Cp / V;
        const dim3 grid((unsigned)(nrows < 4096 ? nrows : 4096), (unsigned)((cpp + 255) / 256));
        if (dtype == YDL_F32) resize_acc_sums_kernel<float, false><<<grid, 256, 0, st>>>(mode, (const float*)x, ldx, (float*)y, ldy, Hi, Wi, Ho, Wo, Cp, sh, sw, nrows, nullptr, 0);
        else resize_acc_sums_kernel<bf16_t, false><<<grid, 256, 0, st>>>(mode, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, Hi, Wi, Ho, Wo, Cp, sh, sw, nrows, nullptr, 0);
        YDL_LAUNCH_CHECK();
        return 0;
    }
    int grid = sgrid((long long)N * Ho * Wo * (Cp / V));
    if (dtype == YDL_F32) resize_fwd_kernel<float><<<grid, 256, 0, st>>>(mode, (const float*)x, ldx, (float*)y, ldy, N, Hi, Wi, Ho, Wo, Cp, sh, sw);
    else resize_fwd_kernel<bf16_t><<<grid, 256, 0, st>>>(mode, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, N, Hi, Wi, Ho, Wo, Cp, sh, sw);
    YDL_LAUNCH_CHECK();
    return 0;
}
extern "C" int ydl_resize_acc_sums(int dtype, int mode, const void* x, int ldx, void* y, int ldy, int N, int Hi, int Wi, int Ho, int Wo,
                                   int C, float scale_h, float scale_w, float* sums, int sums_ld, void* stream) {
    const int V = dtype == YDL_F32 ? 4 : 8;
    const int Cp = round_up(C, V);
    YDL_CHECK(x && y && mode >= 0 && mode <= 2 && ldx >= Cp && ldy >= Cp && N >= 1, "bad arguments");
    YDL_CHECK(aligned16(x) && aligned16(y) && ldx % V == 0 && ldy % V == 0, "16-byte alignment");
    YDL_CHECK(sums == nullptr || sums_ld >= Cp, "statistics rows shorter than the padded channel count");
    const float sh = axis_scale(mode, Hi, Ho, scale_h), sw = axis_scale(mode, Wi, Wo, scale_w);
    hipStream_t st = (hipStream_t)stream;
    const int cpp = Cp / V;
    YDL_CHECK((long long)N * Ho < (1ll << 30), "too many output rows");
    const int nrows = N * Ho;
    const dim3 grid((unsigned)(nrows < 4096 ? nrows : 4096), (unsigned)((cpp + 255) / 256));
    if (dtype == YDL_F32) resize_acc_sums_kernel<float, true><<<grid, 256, 0, st>>>(mode, (const float*)x, ldx, (float*)y, ldy, Hi, Wi, Ho, Wo, Cp, sh, sw, nrows, sums, sums_ld);
    else resize_acc_sums_kernel<bf16_t, true><<<grid, 256, 0, st>>>(mode, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, Hi, Wi, Ho, Wo, Cp, sh, sw, nrows, sums, sums_ld);
    YDL_LAUNCH_CHECK();
    return 0;
}
extern "C" int ydl_resize_bwd(int dtype, int mode, const void* dy, int lddy, void* dx, int lddx, int accumulate,
                              int N, int Hi, int Wi, int Ho, int Wo, int C, float scale_h, float scale_w, void* stream) {
    const int V = dtype == YDL_F32 ? 4 : 8;
    const int Cp = round_up(C, V);
    YDL_CHECK(dy && dx && mode >= 0 && mode <= 2 && lddy >= Cp && lddx >= Cp, "bad arguments");
    float sh = axis_scale(mode, Hi, Ho, scale_h), sw = axis_scale(mode, Wi, Wo, scale_w);
    hipStream_t st = (hipStream_t)stream;
    int grid = sgrid((long long)N * Hi * Wi * (Cp / V));
    // (the fixed 2S-candidate window of the integer-factor kernel holds every referencing output only for the EXACT scale 1/S: a
    // caller-given scale_factor such as 2.05 still yields Ho = 2 Hi by flooring, but maps outputs outside that window — generic path)
    const int Sq = (Wi > 0 && Wo % Wi == 0) ? Wo / Wi : 0;
    if (g_resize_int && mode == 1 && (Sq == 2 || Sq == 4) && Ho == Hi * Sq && sh == 1.f / (float)Sq && sw == 1.f / (float)Sq) {
        const int S = Sq;
        // (the accumulation order over the candidates equals the generic kernel's: rows outer, columns inner, ascending)
        if (dtype == YDL_F32) {
            if (S == 2) resize_bwd_int_kernel<float, 2><<<grid, 256, 0, st>>>((const float*)dy, lddy, (float*)dx, lddx, accumulate, N, Hi, Wi, Cp, sh, sw);
            else resize_bwd_int_kernel<float, 4><<<grid, 256, 0, st>>>((const float*)dy, lddy, (float*)dx, lddx, accumulate, N, Hi, Wi, Cp, sh, sw);
        } else {
            if (S == 2) resize_bwd_int_kernel<bf16_t, 2><<<grid, 256, 0, st>>>((const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, accumulate, N, Hi, Wi, Cp, sh, sw);
            else resize_bwd_int_kernel<bf16_t, 4><<<grid, 256, 0, st>>>((const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, accumulate, N, Hi, Wi, Cp, sh, sw);
        }
        YDL_LAUNCH_CHECK();
        return 0;
    }
    if (dtype == YDL_F32) resize_bwd_kernel<float><<<grid, 256, 0, st>>>(mode, (const float*)dy, lddy, (float*)dx, lddx, accumulate, N, Hi, Wi, Ho, Wo, Cp, sh, sw);
    else resize_bwd_kernel<bf16_t><<<grid, 256, 0, st>>>(mode, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, accumulate, N, Hi, Wi, Ho, Wo, Cp, sh, sw);
    YDL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// strided slice copy / add
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void copy2d_kernel(const T* __restrict__ src, int lds_, T* __restrict__ dst, int ldd,
                                                     long long npix, int Cp, int accumulate) {
    constexpr int V = ET<T>::V;
    const int cpp = Cp / V;
    const long long total = npix * cpp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long pix = i / cpp;
        int c = (int)(i - pix * cpp) * V;
        uint4 v = *(const uint4*)(src + pix * lds_ + c);
        if (accumulate) {
            float a[V], b[V];
            unpack16<T>(v, a);
            unpack16<T>(*(const uint4*)(dst + pix * ldd + c), b);
#pragma unroll
            for (int e = 0; e < V; ++e) a[e] += b[e];
            v = pack16<T>(a);
        }
        *(uint4*)(dst + pix * ldd + c) = v;
    }
}
// element-granular fallback for channel slices that are not 16-byte aligned (c0 or C not a chunk multiple)
template <typename T>
__global__ __launch_bounds__(256) void copy2d_scalar_kernel(const T* __restrict__ src, int lds_, T* __restrict__ dst, int ldd,
                                                            long long npix, int C, int accumulate) {
    const long long total = npix * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long pix = i / C;
        int c = (int)(i - pix * C);
        float v = ET<T>::ld(src + pix * lds_ + c);
        if (accumulate) v += ET<T>::ld(dst + pix * ldd + c);
        ET<T>::st(dst + pix * ldd + c, v);
    }
}
extern "C" int ydl_copy2d(int dtype, const void* src, int lds_, void* dst, int ldd, int64_t npix, int C, int accumulate, void* stream) {
    const int V = dtype == YDL_F32 ? 4 : 8;
    const int Cp = round_up(C, V);
    YDL_CHECK(src && dst && lds_ >= C && ldd >= C, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (C % V != 0 || lds_ % V != 0 || ldd % V != 0 || !aligned16(src) || !aligned16(dst)) {
        int grid = sgrid(npix * C);
        if (dtype == YDL_F32) copy2d_scalar_kernel<float><<<grid, 256, 0, st>>>((const float*)src, lds_, (float*)dst, ldd, npix, C, accumulate);
        else copy2d_scalar_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)src, lds_, (bf16_t*)dst, ldd, npix, C, accumulate);
        YDL_LAUNCH_CHECK();
        return 0;
    }
    int grid = sgrid(npix * (Cp / V));
    if (dtype == YDL_F32) copy2d_kernel<float><<<grid, 256, 0, st>>>((const float*)src, lds_, (float*)dst, ldd, npix, Cp, accumulate);
    else copy2d_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)src, lds_, (bf16_t*)dst, ldd, npix, Cp, accumulate);
    YDL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// model edge: NCHW f32 <-> NHWC T through an LDS transpose tile (64 pixels x C channels)
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int ldd,
                                                           int C, long long HW) {
    // thread = one pixel: coalesced plane reads (lanes = consecutive pixels), one 16-byte NHWC store per chunk
    constexpr int V = ET<T>::V;
    const int n = blockIdx.y;
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float* s = src + (size_t)n * C * HW + p;
    T* d = dst + ((size_t)n * HW + p) * ldd;
    for (int c0 = 0; c0 < ldd; c0 += V) {
        float v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = (c0 + e < C) ? s[(size_t)(c0 + e) * HW] : 0.f;
        *(uint4*)(d + c0) = pack16<T>(v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ src, int lds_, float* __restrict__ dst,
                                                           int C, long long HW, int accumulate) {
    constexpr int V = ET<T>::V;
    const int n = blockIdx.y;
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const T* s = src + ((size_t)n * HW + p) * lds_;
    float* d = dst + (size_t)n * C * HW + p;
    for (int c0 = 0; c0 < C; c0 += V) {
        float v[V];
        unpack16<T>(*(const uint4*)(s + c0), v);
#pragma unroll
        for (int e = 0; e < V; ++e)
            if (c0 + e < C) {
                float o = v[e];
                if (accumulate) o += d[(size_t)(c0 + e) * HW];
                d[(size_t)(c0 + e) * HW] = o;
            }
    }
}
extern "C" int ydl_nchw_to_nhwc(int dtype, const float* src, void* dst, int ldd, int N, int C, int H, int W, void* stream) {
    YDL_CHECK(src && dst && ldd >= C && ldd % 8 == 0, "ldd must cover C and be a multiple of 8");
    long long HW = (long long)H * W;
    dim3 grid((unsigned)((HW + 255) / 256), N);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YDL_F32) nchw_to_nhwc_kernel<float><<<grid, 256, 0, st>>>(src, (float*)dst, ldd, C, HW);
    else nchw_to_nhwc_kernel<bf16_t><<<grid, 256, 0, st>>>(src, (bf16_t*)dst, ldd, C, HW);
    YDL_LAUNCH_CHECK();
    return 0;
}
// Space-to-depth edge conversion for a stem convolution whose kernel size and padding are multiples of its stride s:
//   conv(k, s, p) on (H, W, C)  ==  conv(k/s, 1, p/s) on (H/s, W/s, s*s*C)   with channel (dy*s+dx)*C + c = pixel (s*h+dy, s*w+dx)
// (the 6x6/stride-2 stem of the yolov5 backbone becomes a 3x3/stride-1 conv over 12 channels: K = 9*16 instead of the
// 36*8 of the channel-padded form, and its loader reads 32-byte chunks).  dst is NHWC with pixel stride ldd.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_s2d_kernel(const float* __restrict__ src, T* __restrict__ dst, int ldd,
                                                          int C, int H, int W, int s) {
    const int n = blockIdx.y;
    const int Ho = H / s, Wo = W / s;
    const long long HWo = (long long)Ho * Wo;
    const int Cs = C * s * s;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HWo; i += (long long)gridDim.x * blockDim.x) {
        const int wo = (int)(i % Wo), ho = (int)(i / Wo);
        T* d = dst + ((size_t)n * HWo + i) * ldd;
        constexpr int V = ET<T>::V;
        for (int c0 = 0; c0 < ldd; c0 += V) {            // one 16-byte store per chunk of V channels
            float f[V];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int ch = c0 + e;
                const int c = ch % C, dd = ch / C, dx = dd % s, dy = dd / s;
                f[e] = ch < Cs ? src[(((size_t)n * C + c) * H + (ho * s + dy)) * W + (wo * s + dx)] : 0.f;
            }
            *(uint4*)(d + c0) = pack16<T>(f);
        }
    }
}
// s = 2, C <= 4 (the RGB stem): the two columns of a pixel pair arrive as ONE 8-byte load per (channel, row) — six coalesced loads per
// thread instead of twelve strided ones — and the channel index arithmetic is compile-time (the generic kernel divides by the
// run-time C and s per element: 46 us for the 79 MB input of BASELINE config 2)
template <typename T, int C>
__global__ __launch_bounds__(256) void nchw_to_s2d2_kernel(const float* __restrict__ src, T* __restrict__ dst, int ldd, int H, int W) {
    constexpr int V = ET<T>::V;
    constexpr int Cs = C * 4;
    static_assert(Cs <= 16, "at most 16 space-to-depth channels");
    const int n = blockIdx.y;
    const int Ho = H / 2, Wo = W / 2;
    const long long HWo = (long long)Ho * Wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < HWo; i += (long long)gridDim.x * blockDim.x) {
        const int wo = (int)(i % Wo), ho = (int)(i / Wo);
        float f[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) f[e] = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const float2 v = *(const float2*)(src + (((size_t)n * C + c) * H + (ho * 2 + dy)) * W + wo * 2);
                f[(dy * 2 + 0) * C + c] = v.x;
                f[(dy * 2 + 1) * C + c] = v.y;
            }
        T* d = dst + ((size_t)n * HWo + i) * ldd;
        for (int c0 = 0; c0 < ldd; c0 += V) {
            float g[V];
#pragma unroll
            for (int e = 0; e < V; ++e) g[e] = 0.f;
            if (c0 < 16) {
#pragma unroll
                for (int e = 0; e < V; ++e) g[e] = c0 == 0 ? f[e] : (c0 == 4 ? f[(4 + e) & 15] : (c0 == 8 ? f[(8 + e) & 15] : f[(12 + e) & 15]));
            }
            *(uint4*)(d + c0) = pack16<T>(g);
        }
    }
}

extern "C" int ydl_nchw_to_s2d(int dtype, const float* src, void* dst, int ldd, int N, int C, int H, int W, int s, void* stream) {
    YDL_CHECK(src && dst && s >= 1 && H % s == 0 && W % s == 0 && ldd >= C * s * s && ldd % 8 == 0,
              "H and W must be multiples of s; ldd must cover C*s*s and be a multiple of 8");
    long long HWo = (long long)(H / s) * (W / s);
    dim3 grid((unsigned)((HWo + 255) / 256), N);
    hipStream_t st = (hipStream_t)stream;
    if (s == 2 && C == 3 && W % 2 == 0 && ((uintptr_t)src & 7) == 0) {
        if (dtype == YDL_F32) nchw_to_s2d2_kernel<float, 3><<<grid, 256, 0, st>>>(src, (float*)dst, ldd, H, W);
        else nchw_to_s2d2_kernel<bf16_t, 3><<<grid, 256, 0, st>>>(src, (bf16_t*)dst, ldd, H, W);
        YDL_LAUNCH_CHECK();
        return 0;
    }
    if (dtype == YDL_F32) nchw_to_s2d_kernel<float><<<grid, 256, 0, st>>>(src, (float*)dst, ldd, C, H, W, s);
    else nchw_to_s2d_kernel<bf16_t><<<grid, 256, 0, st>>>(src, (bf16_t*)dst, ldd, C, H, W, s);
    YDL_LAUNCH_CHECK();
    return 0;
}
extern "C" int ydl_nhwc_to_nchw(int dtype, const void* src, int lds_, float* dst, int N, int C, int H, int W, int accumulate, void* stream) {
    YDL_CHECK(src && dst && lds_ >= round_up(C, dtype == YDL_F32 ? 4 : 8), "source stride must cover C rounded to a chunk");
    long long HW = (long long)H * W;
    dim3 grid((unsigned)((HW + 255) / 256), N);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YDL_F32) nhwc_to_nchw_kernel<float><<<grid, 256, 0, st>>>((const float*)src, lds_, dst, C, HW, accumulate);
    else nhwc_to_nchw_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)src, lds_, dst, C, HW, accumulate);
    YDL_LAUNCH_CHECK();
    return 0;
}

// y[n,p,c] = x[n,p,c] * gate[n,c]
template <typename T>
__global__ __launch_bounds__(256) void scale_channels_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ gate,
                                                             T* __restrict__ y, int ldy, int N, long long hw, int C, int Cp) {
    constexpr int V = ET<T>::V;
    const int cpp = Cp / V;
    const long long total = (long long)N * hw * cpp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long pix = i / cpp;
        int c = (int)(i - pix * cpp) * V;
        int n = (int)(pix / hw);
        float v[V];
        unpack16<T>(*(const uint4*)(x + pix * ldx + c), v);
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] *= (c + e < C) ? gate[(size_t)n * C + c + e] : 0.f;
        *(uint4*)(y + pix * ldy + c) = pack16<T>(v);
    }
}
extern "C" int ydl_scale_channels(int dtype, const void* x, int ldx, const float* gate, void* y, int ldy,
                                  int N, int64_t hw, int C, void* stream) {
    const int V = dtype == YDL_F32 ? 4 : 8;
    const int Cp = round_up(C, V);
    YDL_CHECK(x && y && gate && ldx >= Cp && ldy >= Cp, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int grid = sgrid((long long)N * hw * (Cp / V));
    if (dtype == YDL_F32) scale_channels_kernel<float><<<grid, 256, 0, st>>>((const float*)x, ldx, gate, (float*)y, ldy, N, hw, C, Cp);
    else scale_channels_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, ldx, gate, (bf16_t*)y, ldy, N, hw, C, Cp);
    YDL_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------------------------------------------
// GAM support (unet-lite/yolo9-seg/seg_diceloss_yolov9.py:475-510): global avg / max pooling to 1x1, the sigmoid
// gate, per-(n,c) dot product for the gate gradient.  One CTA per (image, 16-byte channel chunk).
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void global_pool_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ avg, int lda,
                                                              T* __restrict__ mx, int ldm, int* __restrict__ amax,
                                                              long long HW, int Cp) {
    constexpr int V = ET<T>::V;
    const int n = blockIdx.y, c = blockIdx.x * V;
    const T* xb = x + (size_t)n * HW * ldx + c;
    float s[V], m[V];
    int mi[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { s[e] = 0.f; m[e] = -INFINITY; mi[e] = 0; }
    for (long long p = threadIdx.x; p < HW; p += 256) {
        float v[V];
        unpack16<T>(*(const uint4*)(xb + p * ldx), v);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            s[e] += v[e];
            if (v[e] > m[e] || v[e] != v[e]) { m[e] = v[e]; mi[e] = (int)p; }
        }
    }
    __shared__ float ss[256][8], sm[256][8];
    __shared__ int si[256][8];
#pragma unroll
    for (int e = 0; e < V; ++e) { ss[threadIdx.x][e] = s[e]; sm[threadIdx.x][e] = m[e]; si[threadIdx.x][e] = mi[e]; }
    __syncthreads();
    if (threadIdx.x < V) {
        const int e = threadIdx.x;
        float a = 0.f, b = -INFINITY;
        int bi = 0;
        for (int t = 0; t < 256; ++t) {
            a += ss[t][e];
            float v = sm[t][e];
            int vi = si[t][e];
            // first maximum in scan order (ATen adaptive_max_pool semantics): larger value, or equal value at lower index
            if (v > b || (v == b && vi < bi)) { b = v; bi = vi; }
        }
        ET<T>::st(avg + (size_t)n * lda + c + e, a / (float)HW);
        ET<T>::st(mx + (size_t)n * ldm + c + e, b);
        amax[(size_t)n * Cp + c + e] = bi;
    }
}
template <typename T>
__global__ __launch_bounds__(256) void global_pool_bwd_kernel(const T* __restrict__ davg, int lda, const T* __restrict__ dmx, int ldm,
                                                              const int* __restrict__ amax, T* __restrict__ dx, int lddx,
                                                              int accumulate, int N, long long HW, int Cp) {
    constexpr int V = ET<T>::V;
    const int cpp = Cp / V;
    const long long total = (long long)N * HW * cpp;
    const float inv = 1.f / (float)HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int cq = (int)(i % cpp);
        long long pix = i / cpp;
        int n = (int)(pix / HW);
        int p = (int)(pix - (long long)n * HW);
        int c = cq * V;
        float g[V];
#pragma unroll
        for (int e = 0; e < V; ++e) g[e] = 0.f;
        if (accumulate) unpack16<T>(*(const uint4*)(dx + pix * lddx + c), g);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            if (davg) g[e] += ET<T>::ld(davg + (size_t)n * lda + c + e) * inv;
            if (dmx && amax[(size_t)n * Cp + c + e] == p) g[e] += ET<T>::ld(dmx + (size_t)n * ldm + c + e);
        }
        *(uint4*)(dx + pix * lddx + c) = pack16<T>(g);
    }
}
extern "C" int ydl_global_pool_fwd(int dtype, const void* x, int ldx, void* avg, int lda, void* mx, int ldm, int32_t* argmax,
                                   int N, int64_t HW, int C, void* stream) {
    const int V = dtype == YDL_F32 ? 4 : 8;
    const int Cp = round_up(C, V);
    YDL_CHECK(x && avg && mx && argmax && ldx >= Cp && lda >= Cp && ldm >= Cp, "bad arguments");
    dim3 grid(Cp / V, N);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YDL_F32) global_pool_fwd_kernel<float><<<grid, 256, 0, st>>>((const float*)x, ldx, (float*)avg, lda, (float*)mx, ldm, argmax, HW, Cp);
    else global_pool_fwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, ldx, (bf16_t*)avg, lda, (bf16_t*)mx, ldm, argmax, HW, Cp);
    YDL_LAUNCH_CHECK();
    return 0;
}
extern "C" int ydl_global_pool_bwd(int dtype, const void* davg, int lda, const void* dmx, int ldm, const int32_t* argmax,
                                   void* dx, int lddx, int accumulate, int N, int64_t HW, int C, void* stream) {
    const int V = dtype == YDL_F32 ? 4 : 8;
    const int Cp = round_up(C, V);
    YDL_CHECK(dx && argmax && lddx >= Cp && (davg || dmx), "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int grid = sgrid((long long)N * HW * (Cp / V));
    if (dtype == YDL_F32) global_pool_bwd_kernel<float><<<grid, 256, 0, st>>>((const float*)davg, lda, (const float*)dmx, ldm, argmax, (float*)dx, lddx, accumulate, N, HW, Cp);
    else global_pool_bwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)davg, lda, (const bf16_t*)dmx, ldm, argmax, (bf16_t*)dx, lddx, accumulate, N, HW, Cp);
    YDL_LAUNCH_CHECK();
    return 0;
}

// gate[n][c] = sigmoid(a[n][c] + b[n][c]);   backward: da (+)= dgate*g*(1-g), db likewise
template <typename T>
__global__ void gate_fwd_kernel(const T* a, int lda, const T* b, int ldb, float* gate, int N, int C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N * C) {
        int n = i / C, c = i % C;
        gate[i] = sigmoid_f(ET<T>::ld(a + (size_t)n * lda + c) + ET<T>::ld(b + (size_t)n * ldb + c));
    }
}
template <typename T>
__global__ void gate_bwd_kernel(const float* gate, const float* dgate, T* da, int lda, int acc_a, T* db, int ldb, int acc_b, int N, int C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N * C) {
        int n = i / C, c = i % C;
        float g = gate[i];
        float d = dgate[i] * g * (1.f - g);
        T* pa = da + (size_t)n * lda + c;
        T* pb = db + (size_t)n * ldb + c;
        ET<T>::st(pa, acc_a ? ET<T>::ld(pa) + d : d);
        ET<T>::st(pb, acc_b ? ET<T>::ld(pb) + d : d);
    }
}
extern "C" int ydl_gate_fwd(int dtype, const void* a, int lda, const void* b, int ldb, float* gate, int N, int C, void* stream) {
    YDL_CHECK(a && b && gate, "null pointer");
    int grid = (N * C + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YDL_F32) gate_fwd_kernel<float><<<grid, 256, 0, st>>>((const float*)a, lda, (const float*)b, ldb, gate, N, C);
    else gate_fwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)a, lda, (const bf16_t*)b, ldb, gate, N, C);
    YDL_LAUNCH_CHECK();
    return 0;
}
extern "C" int ydl_gate_bwd(int dtype, const float* gate, const float* dgate, void* da, int lda, int acc_a, void* db, int ldb,
                            int acc_b, int N, int C, void* stream) {
    YDL_CHECK(gate && dgate && da && db, "null pointer");
    int grid = (N * C + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YDL_F32) gate_bwd_kernel<float><<<grid, 256, 0, st>>>(gate, dgate, (float*)da, lda, acc_a, (float*)db, ldb, acc_b, N, C);
    else gate_bwd_kernel<bf16_t><<<grid, 256, 0, st>>>(gate, dgate, (bf16_t*)da, lda, acc_a, (bf16_t*)db, ldb, acc_b, N, C);
    YDL_LAUNCH_CHECK();
    return 0;
}

// out[n][c] = sum_p a[n,p,c] * b[n,p,c]
template <typename T>
__global__ __launch_bounds__(256) void channel_dot_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb,
                                                          float* __restrict__ out, long long HW, int C) {
    constexpr int V = ET<T>::V;
    const int n = blockIdx.y, c = blockIdx.x * V;
    float s[V];
#pragma unroll
    for (int e = 0; e < V; ++e) s[e] = 0.f;
    for (long long p = threadIdx.x; p < HW; p += 256) {
        float x[V], y[V];
        unpack16<T>(*(const uint4*)(a + ((size_t)n * HW + p) * lda + c), x);
        unpack16<T>(*(const uint4*)(b + ((size_t)n * HW + p) * ldb + c), y);
#pragma unroll
        for (int e = 0; e < V; ++e) s[e] += x[e] * y[e];
    }
    __shared__ float ss[4][8];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        float v = wave_sum(s[e]);
        if ((threadIdx.x & 63) == 0) ss[threadIdx.x >> 6][e] = v;
    }
    __syncthreads();
    if (threadIdx.x < V && c + threadIdx.x < C)
        out[(size_t)n * C + c + threadIdx.x] = ss[0][threadIdx.x] + ss[1][threadIdx.x] + ss[2][threadIdx.x] + ss[3][threadIdx.x];
}
extern "C" int ydl_channel_dot(int dtype, const void* a, int lda, const void* b, int ldb, float* out, int N, int64_t HW, int C,
                               void* stream) {
    const int V = dtype == YDL_F32 ? 4 : 8;
    const int Cp = round_up(C, V);
    YDL_CHECK(a && b && out && lda >= Cp && ldb >= Cp, "bad arguments");
    dim3 grid(Cp / V, N);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YDL_F32) channel_dot_kernel<float><<<grid, 256, 0, st>>>((const float*)a, lda, (const float*)b, ldb, out, HW, C);
    else channel_dot_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)a, lda, (const bf16_t*)b, ldb, out, HW, C);
    YDL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// zero fills (the taped region issues no ATen kernel; these are recordable entry points like every other launch)
// ------------------------------------------------------------------------------------------------------
// a plain kernel, not hipMemsetAsync: a memset becomes a memset NODE under HIP-graph capture, and a chain of captured graph
// segments with memset nodes in it replayed with wrong contents on ROCm 7.2 (tests/test_gpu_dp.py, segmented capture); as a kernel
// the fill is ordered like every other launch, eagerly, in a graph and in a launch list
__global__ __launch_bounds__(256) void fill_zero_kernel(unsigned char* __restrict__ dst, long long bytes) {
    const long long head = (16 - ((uintptr_t)dst & 15)) & 15;        // bytes up to the first 16-byte boundary
    const long long h = head < bytes ? head : bytes;
    const long long nvec = (bytes - h) >> 4;
    uint4* v = (uint4*)(dst + h);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x)
        v[i] = make_uint4(0, 0, 0, 0);
    if (blockIdx.x == 0) {
        for (long long i = threadIdx.x; i < h; i += blockDim.x) dst[i] = 0;
        const long long tail0 = h + (nvec << 4);
        for (long long i = tail0 + threadIdx.x; i < bytes; i += blockDim.x) dst[i] = 0;
    }
}
extern "C" int ydl_fill_zero(void* dst, int64_t bytes, void* stream) {
    YDL_CHECK(dst != nullptr || bytes == 0, "null destination");
    YDL_CHECK(bytes >= 0, "negative size");
    if (bytes == 0) return 0;
    fill_zero_kernel<<<sgrid((bytes + 15) / 16), 256, 0, (hipStream_t)stream>>>((unsigned char*)dst, (long long)bytes);
    YDL_LAUNCH_CHECK();
    return 0;
}
template <typename T>
__global__ __launch_bounds__(256) void zero2d_kernel(T* __restrict__ dst, int ldd, long long npix, int C) {
    const long long total = npix * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long pix = i / C;
        int c = (int)(i - pix * C);
        dst[pix * ldd + c] = (T)0;
    }
}
extern "C" int ydl_zero2d(int dtype, void* dst, int ldd, int64_t npix, int C, void* stream) {
    YDL_CHECK(dst && ldd >= C && C > 0 && npix >= 0, "bad arguments");
    if (npix == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (ldd == C) return ydl_fill_zero(dst, npix * C * (int64_t)esize(dtype), stream);
    int grid = sgrid(npix * C);
    if (dtype == YDL_F32) zero2d_kernel<float><<<grid, 256, 0, st>>>((float*)dst, ldd, npix, C);
    else zero2d_kernel<unsigned short><<<grid, 256, 0, st>>>((unsigned short*)dst, ldd, npix, C);
    YDL_LAUNCH_CHECK();
    return 0;
}
