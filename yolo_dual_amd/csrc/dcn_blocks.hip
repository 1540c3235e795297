// Pieces of the DCNv3 *module* (models/ops_dcnv3/build/.../modules/dcnv3.py:50-136) around the deformable-sampling op:
//   * depth-wise k x k convolution (the `dw_conv = Conv(c, c, k, g=c)` branch, :89) forward / input gradient / weight gradient
//   * generic per-block BatchNorm partial statistics of an NHWC tensor (same [nblocks][2][C] (sum, M2) contract as the conv
//     epilogue, so ydl_bn_finalize / ydl_bn_act_fwd / ydl_bn_act_bwd serve the depth-wise branch unchanged)
//   * per-channel sums over pixels (bias gradient of the NHWC `nn.Linear` layers = 1x1 convolutions with bias)
//   * soft-max over the K*K sampling points of each group (`F.softmax(mask.reshape(N,H,W,G,-1), -1)`, :122-123)
//   * f32 -> compute-dtype cast with optional accumulation (the op returns f32 gradients, dcnv3_cuda.cu:126-133)
// All HBM-bound: a thread owns one 16-byte channel chunk, pixels are walked with a grid stride.
#include "common.h"

#define DW_MAXTAPS 49

// ------------------------------------------------------------------------------------------------------
// depth-wise convolution, stride 1, padding p, NHWC.  w: f32 [C][k*k] (nn.Conv2d(C, C, k, groups=C).weight is [C,1,k,k])
// FLIP = false: y[n,h,w,c] = sum_{r,s} w[c][r*k+s] * x[n, h+r-p, w+s-p, c]            (forward)
// FLIP = true : dx[n,h,w,c] = sum_{r,s} w[c][r*k+s] * dy[n, h-r+p, w-s+p, c]          (input gradient)
// ------------------------------------------------------------------------------------------------------
template <typename T, bool FLIP>
__global__ __launch_bounds__(256) void dwconv_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ w, T* __restrict__ y,
                                                     int ldy, int accumulate, int N, int H, int W, int Cp, int C, int k, int p) {
    constexpr int V = ET<T>::V;
    const int cpp = Cp / V;
    const long long total = (long long)N * H * W * cpp;
    const int kk = k * k;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cq = (int)(i % cpp);
        const long long pix = i / cpp;
        const int wx = (int)(pix % W);
        const long long t2 = pix / W;
        const int hy = (int)(t2 % H);
        const int n = (int)(t2 / H);
        const int c0 = cq * V;
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int r = 0; r < k; ++r) {
            const int ih = FLIP ? hy - r + p : hy + r - p;
            if ((unsigned)ih >= (unsigned)H) continue;
            for (int s = 0; s < k; ++s) {
                const int iw = FLIP ? wx - s + p : wx + s - p;
                if ((unsigned)iw >= (unsigned)W) continue;
                float v[V];
                unpack16<T>(*(const uint4*)(x + ((size_t)(n * H + ih) * W + iw) * ldx + c0), v);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float wv = (c0 + e) < C ? w[(size_t)(c0 + e) * kk + r * k + s] : 0.f;
                    acc[e] = fmaf(wv, v[e], acc[e]);
                }
            }
        }
        T* dst = y + (size_t)pix * ldy + c0;
        if (accumulate) {
            float o[V];
            unpack16<T>(*(const uint4*)dst, o);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += o[e];
        }
        *(uint4*)dst = pack16<T>(acc);
    }
}

static inline int stream_grid(long long work_items) {
    long long b = (work_items + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

static int dw_check(int dtype, const void* a, const void* b, const void* c, int lda, int ldb, int C, int k, int p) {
    YDL_CHECK(dtype == YDL_F32 || dtype == YDL_BF16, "bad dtype");
    YDL_CHECK(a && b && c, "null pointer");
    YDL_CHECK(C > 0 && (k == 1 || k == 3 || k == 5 || k == 7) && 2 * p == k - 1, "depth-wise conv: k in {1,3,5,7} with 'same' padding, stride 1");
    const int V = dtype == YDL_F32 ? 4 : 8;
    YDL_CHECK(lda >= round_up(C, V) && ldb >= round_up(C, V), "pixel strides must cover C rounded up to a 16-byte chunk");
    YDL_CHECK(aligned16(a) && aligned16(c), "16-byte alignment");
    return 0;
}

extern "C" int ydl_dwconv_fwd(int dtype, const void* x, int ldx, const float* w, void* y, int ldy, int N, int H, int W, int C,
                              int k, int p, void* stream) {
    if (int e = dw_check(dtype, x, w, y, ldx, ldy, C, k, p)) return e;
    const int V = dtype == YDL_F32 ? 4 : 8, Cp = round_up(C, V);
    hipStream_t st = (hipStream_t)stream;
    const int grid = stream_grid((long long)N * H * W * (Cp / V));
    if (dtype == YDL_F32) dwconv_kernel<float, false><<<grid, 256, 0, st>>>((const float*)x, ldx, w, (float*)y, ldy, 0, N, H, W, Cp, C, k, p);
    else dwconv_kernel<bf16_t, false><<<grid, 256, 0, st>>>((const bf16_t*)x, ldx, w, (bf16_t*)y, ldy, 0, N, H, W, Cp, C, k, p);
    YDL_LAUNCH_CHECK();
    return 0;
}

extern "C" int ydl_dwconv_dgrad(int dtype, const void* dy, int lddy, const float* w, void* dx, int lddx, int accumulate, int N, int H,
                                int W, int C, int k, int p, void* stream) {
    if (int e = dw_check(dtype, dy, w, dx, lddy, lddx, C, k, p)) return e;
    const int V = dtype == YDL_F32 ? 4 : 8, Cp = round_up(C, V);
    hipStream_t st = (hipStream_t)stream;
    const int grid = stream_grid((long long)N * H * W * (Cp / V));
    if (dtype == YDL_F32) dwconv_kernel<float, true><<<grid, 256, 0, st>>>((const float*)dy, lddy, w, (float*)dx, lddx, accumulate, N, H, W, Cp, C, k, p);
    else dwconv_kernel<bf16_t, true><<<grid, 256, 0, st>>>((const bf16_t*)dy, lddy, w, (bf16_t*)dx, lddx, accumulate, N, H, W, Cp, C, k, p);
    YDL_LAUNCH_CHECK();
    return 0;
}

// weight gradient: dw[c][r*k+s] += sum_pixels dy[n,h,w,c] * x[n, h+r-p, w+s-p, c].  Deterministic two stages: CTA b sums its
// pixel range into part[b][c][tap] (thread = channel, fixed order), then one thread per (c, tap) adds the partials in order.
#define DW_WG_BLOCKS 512
template <typename T, int k>
__global__ __launch_bounds__(256) void dwconv_wgrad_part_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy,
                                                                float* __restrict__ part, int N, int H, int W, int C, int p,
                                                                long long per_block) {
    constexpr int kk = k * k;
    const long long npix = (long long)N * H * W;
    const long long p0 = (long long)blockIdx.x * per_block;
    long long p1 = p0 + per_block;
    if (p1 > npix) p1 = npix;
    for (int c = blockIdx.y * 256 + threadIdx.x; c < C; c += gridDim.y * 256) {
        float acc[kk];
#pragma unroll
        for (int t = 0; t < kk; ++t) acc[t] = 0.f;
        for (long long pix = p0; pix < p1; ++pix) {
            const int wx = (int)(pix % W);
            const long long t2 = pix / W;
            const int hy = (int)(t2 % H);
            const int n = (int)(t2 / H);
            const float g = ET<T>::ld(dy + (size_t)pix * lddy + c);
#pragma unroll
            for (int r = 0; r < k; ++r) {
                const int ih = hy + r - p;
#pragma unroll
                for (int s = 0; s < k; ++s) {
                    const int iw = wx + s - p;
                    const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                    const float xv = ok ? ET<T>::ld(x + ((size_t)(n * H + ih) * W + iw) * ldx + c) : 0.f;
                    acc[r * k + s] = fmaf(g, xv, acc[r * k + s]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < kk; ++t) part[((size_t)blockIdx.x * C + c) * kk + t] = acc[t];
    }
}
// k = 3, 16-byte-aligned rows: a thread owns one 16-byte channel chunk (V channels x 9 taps = 72 accumulators for bf16), the CTA walks
// pixels R at a time with a grid stride; the R pixel lanes are folded through LDS one tap at a time.  (The scalar kernel above keeps
// one channel per thread and a serial pixel loop: 3.2 ms per step of the DCNv3 yolov9 model, against 0.3 ms here.)
template <typename T>
__global__ __launch_bounds__(256) void dwconv3_wgrad_vec_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ dy, int lddy,
                                                                float* __restrict__ part, int N, int H, int W, int C, int Cp) {
    constexpr int V = ET<T>::V;
    const int cpp = Cp / V;
    const int cpb = cpp < 256 ? cpp : 256;
    const int R = 256 / cpb;
    const int cq = threadIdx.x % cpb, pl = threadIdx.x / cpb;
    const int chunk = blockIdx.y * 256 + cq;
    const bool live = pl < R && chunk < cpp;
    const int c0 = chunk * V;
    const long long npix = (long long)N * H * W;
    float acc[9][V];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < V; ++e) acc[t][e] = 0.f;
    if (live) {
        for (long long pix = (long long)blockIdx.x * R + pl; pix < npix; pix += (long long)gridDim.x * R) {
            const int wx = (int)(pix % W);
            const long long t2 = pix / W;
            const int hy = (int)(t2 % H);
            const int n = (int)(t2 / H);
            float g[V];
            unpack16<T>(*(const uint4*)(dy + (size_t)pix * lddy + c0), g);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int ih = hy + r - 1;
                const bool vy = (unsigned)ih < (unsigned)H;
                const int ihc = vy ? ih : hy;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int iw = wx + q - 1;
                    const bool ok = vy && (unsigned)iw < (unsigned)W;
                    const int iwc = (unsigned)iw < (unsigned)W ? iw : wx;
                    float xv[V];
                    unpack16<T>(*(const uint4*)(x + ((size_t)(n * H + ihc) * W + iwc) * ldx + c0), xv);
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[r * 3 + q][e] = fmaf(g[e], ok ? xv[e] : 0.f, acc[r * 3 + q][e]);
                }
            }
        }
    }
    __shared__ float red[256 * 8];
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < V; ++e) red[threadIdx.x * V + e] = acc[t][e];
        __syncthreads();
        if (threadIdx.x < cpb && blockIdx.y * 256 + threadIdx.x < cpp) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float a = 0.f;
                for (int l = 0; l < R; ++l) a += red[(l * cpb + threadIdx.x) * V + e];
                const int c = c0 + e;            // (pl == 0 here: c0 is this thread's own chunk)
                if (c < C) part[((size_t)blockIdx.x * C + c) * 9 + t] = a;
            }
        }
    }
}

__global__ __launch_bounds__(256) void dwconv_wgrad_merge_kernel(const float* __restrict__ part, float* __restrict__ dw, int nblk, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += part[(size_t)b * n + i];
    dw[i] += s;
}
extern "C" int64_t ydl_dwconv_wgrad_ws_bytes(int C, int k) { return (int64_t)DW_WG_BLOCKS * C * k * k * (int64_t)sizeof(float); }
extern "C" int ydl_dwconv_wgrad(int dtype, const void* x, int ldx, const void* dy, int lddy, float* dw, float* ws, int N, int H, int W,
                                int C, int k, int p, void* stream) {
    if (int e = dw_check(dtype, x, dy, dw, ldx, lddy, C, k, p)) return e;
    YDL_CHECK(ws != nullptr, "workspace of ydl_dwconv_wgrad_ws_bytes() required");
    const long long npix = (long long)N * H * W;
    hipStream_t st = (hipStream_t)stream;
    {
        const int V = dtype == YDL_F32 ? 4 : 8, Cp = round_up(C, V);
        static const int novec = getenv("YDL_DW_NOVEC") ? atoi(getenv("YDL_DW_NOVEC")) : 0;
        if (k == 3 && !novec && ldx % V == 0 && lddy % V == 0 && aligned16(x) && aligned16(dy)) {
            const int cpp = Cp / V, cpb = cpp < 256 ? cpp : 256, R = 256 / cpb;
            long long gx = (npix + R - 1) / R;
            if (gx > DW_WG_BLOCKS) gx = DW_WG_BLOCKS;
            dim3 vgrid((unsigned)gx, (unsigned)((cpp + 255) / 256));
            if (dtype == YDL_F32) dwconv3_wgrad_vec_kernel<float><<<vgrid, 256, 0, st>>>((const float*)x, ldx, (const float*)dy, lddy, ws, N, H, W, C, Cp);
            else dwconv3_wgrad_vec_kernel<bf16_t><<<vgrid, 256, 0, st>>>((const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, ws, N, H, W, C, Cp);
            const int n = C * 9;
            dwconv_wgrad_merge_kernel<<<(n + 255) / 256, 256, 0, st>>>(ws, dw, (int)gx, n);
            YDL_LAUNCH_CHECK();
            return 0;
        }
    }
    int nblk = (int)((npix + 63) / 64);
    if (nblk > DW_WG_BLOCKS) nblk = DW_WG_BLOCKS;
    const long long per = (npix + nblk - 1) / nblk;
    nblk = (int)((npix + per - 1) / per);
    dim3 grid(nblk, (C + 255) / 256);
#define DW_WG_LAUNCH(KK)                                                                                                              \
    do {                                                                                                                              \
        if (dtype == YDL_F32)                                                                                                         \
            dwconv_wgrad_part_kernel<float, KK><<<grid, 256, 0, st>>>((const float*)x, ldx, (const float*)dy, lddy, ws, N, H, W, C, p, per);     \
        else                                                                                                                          \
            dwconv_wgrad_part_kernel<bf16_t, KK><<<grid, 256, 0, st>>>((const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, ws, N, H, W, C, p, per); \
    } while (0)
    if (k == 1) DW_WG_LAUNCH(1);
    else if (k == 3) DW_WG_LAUNCH(3);
    else if (k == 5) DW_WG_LAUNCH(5);
    else DW_WG_LAUNCH(7);
    const int n = C * k * k;
    dwconv_wgrad_merge_kernel<<<(n + 255) / 256, 256, 0, st>>>(ws, dw, nblk, n);
    YDL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// BN partial statistics of an arbitrary NHWC tensor: block b covers pixels [b*block_m, (b+1)*block_m): (sum, M2 about the
// block mean) per channel -> part[b][0][c], part[b][1][c]  (row stride round_up(C, 8)): the contract of ydl_bn_finalize.
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ y, int ldy, float* __restrict__ part, long long npix, int C,
                                                       int ldp, int block_m) {
    const long long p0 = (long long)blockIdx.x * block_m;
    long long p1 = p0 + block_m;
    if (p1 > npix) p1 = npix;
    const float inv = 1.f / (float)(p1 - p0);
    for (int c = blockIdx.y * 256 + threadIdx.x; c < C; c += gridDim.y * 256) {
        float s = 0.f;
        for (long long pix = p0; pix < p1; ++pix) s += ET<T>::ld(y + (size_t)pix * ldy + c);
        const float mu = s * inv;
        float q = 0.f;
        for (long long pix = p0; pix < p1; ++pix) {
            const float d = ET<T>::ld(y + (size_t)pix * ldy + c) - mu;
            q = fmaf(d, d, q);
        }
        part[((size_t)blockIdx.x * 2) * ldp + c] = s;
        part[((size_t)blockIdx.x * 2 + 1) * ldp + c] = q;
    }
}
#define BN_STATS_BLOCK_M 64
extern "C" int ydl_bn_stats_block_m(void) { return BN_STATS_BLOCK_M; }
extern "C" int64_t ydl_bn_stats_ws_bytes(int64_t npix, int C) {
    const int64_t nb = (npix + BN_STATS_BLOCK_M - 1) / BN_STATS_BLOCK_M;
    return (nb + nb / 64 + 2) * 2 * round_up(C, 8) * (int64_t)sizeof(float);      // + room for ydl_bn_finalize's level-1 rows
}
extern "C" int ydl_bn_stats(int dtype, const void* y, int ldy, float* ws, int64_t npix, int C, void* stream) {
    YDL_CHECK(dtype == YDL_F32 || dtype == YDL_BF16, "bad dtype");
    YDL_CHECK(y && ws && npix > 0 && C > 0 && ldy >= C, "bad arguments");
    const int nb = (int)((npix + BN_STATS_BLOCK_M - 1) / BN_STATS_BLOCK_M);
    dim3 grid(nb, (C + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YDL_F32) bn_stats_kernel<float><<<grid, 256, 0, st>>>((const float*)y, ldy, ws, npix, C, round_up(C, 8), BN_STATS_BLOCK_M);
    else bn_stats_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)y, ldy, ws, npix, C, round_up(C, 8), BN_STATS_BLOCK_M);
    YDL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// per-channel sum over pixels (bias gradient): out[c] (+)= sum_p x[p][c], deterministic two stages
// ------------------------------------------------------------------------------------------------------
#define CS_BLOCKS 256
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_part_kernel(const T* __restrict__ x, int ldx, float* __restrict__ part, long long npix,
                                                               int C, long long per) {
    const long long p0 = (long long)blockIdx.x * per;
    long long p1 = p0 + per;
    if (p1 > npix) p1 = npix;
    for (int c = blockIdx.y * 256 + threadIdx.x; c < C; c += gridDim.y * 256) {
        float s = 0.f;
        for (long long pix = p0; pix < p1; ++pix) s += ET<T>::ld(x + (size_t)pix * ldx + c);
        part[(size_t)blockIdx.x * C + c] = s;
    }
}
// 16-byte-aligned rows: thread = one 16-byte channel chunk, R pixel lanes per CTA, two pixels in flight per thread
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_vec_kernel(const T* __restrict__ x, int ldx, float* __restrict__ part, long long npix,
                                                              int C, int Cp) {
    constexpr int V = ET<T>::V;
    const int cpp = Cp / V;
    const int cpb = cpp < 256 ? cpp : 256;
    const int R = 256 / cpb;
    const int cq = threadIdx.x % cpb, pl = threadIdx.x / cpb;
    const int chunk = blockIdx.y * 256 + cq;
    const bool live = pl < R && chunk < cpp;
    const int c0 = chunk * V;
    float s[V];
#pragma unroll
    for (int e = 0; e < V; ++e) s[e] = 0.f;
    if (live) {
        const long long stride = (long long)gridDim.x * R;
        long long pix = (long long)blockIdx.x * R + pl;
        for (; pix + stride < npix; pix += 2 * stride) {
            const uint4 q0 = *(const uint4*)(x + (size_t)pix * ldx + c0), q1 = *(const uint4*)(x + (size_t)(pix + stride) * ldx + c0);
            float v0[V], v1[V];
            unpack16<T>(q0, v0);
            unpack16<T>(q1, v1);
#pragma unroll
            for (int e = 0; e < V; ++e) s[e] += v0[e] + v1[e];
        }
        for (; pix < npix; pix += stride) {
            float v0[V];
            unpack16<T>(*(const uint4*)(x + (size_t)pix * ldx + c0), v0);
#pragma unroll
            for (int e = 0; e < V; ++e) s[e] += v0[e];
        }
    }
    __shared__ float red[256 * 8];
#pragma unroll
    for (int e = 0; e < V; ++e) red[threadIdx.x * V + e] = s[e];
    __syncthreads();
    if (threadIdx.x < cpb && chunk < cpp) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float a = 0.f;
            for (int l = 0; l < R; ++l) a += red[(l * cpb + threadIdx.x) * V + e];
            if (c0 + e < C) part[(size_t)blockIdx.x * C + c0 + e] = a;
        }
    }
}
__global__ __launch_bounds__(256) void channel_sum_merge_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk, int C,
                                                                int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)part[(size_t)b * C + c];
    out[c] = (accumulate ? out[c] : 0.f) + (float)s;
}
extern "C" int64_t ydl_channel_sum_ws_bytes(int C) { return (int64_t)CS_BLOCKS * C * (int64_t)sizeof(float); }
extern "C" int ydl_channel_sum(int dtype, const void* x, int ldx, float* out, float* ws, int64_t npix, int C, int accumulate, void* stream) {
    YDL_CHECK(dtype == YDL_F32 || dtype == YDL_BF16, "bad dtype");
    YDL_CHECK(x && out && ws && npix > 0 && C > 0 && ldx >= C, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    {
        const int V = dtype == YDL_F32 ? 4 : 8, Cp = round_up(C, V);
        static const int novec = getenv("YDL_DW_NOVEC") ? atoi(getenv("YDL_DW_NOVEC")) : 0;
        if (!novec && ldx >= Cp && ldx % V == 0 && aligned16(x)) {
            const int cpp = Cp / V, cpb = cpp < 256 ? cpp : 256, R = 256 / cpb;
            long long gx = (npix + 2 * R - 1) / (2 * R);
            if (gx > CS_BLOCKS) gx = CS_BLOCKS;
            if (gx < 1) gx = 1;
            dim3 vgrid((unsigned)gx, (unsigned)((cpp + 255) / 256));
            if (dtype == YDL_F32) channel_sum_vec_kernel<float><<<vgrid, 256, 0, st>>>((const float*)x, ldx, ws, npix, C, Cp);
            else channel_sum_vec_kernel<bf16_t><<<vgrid, 256, 0, st>>>((const bf16_t*)x, ldx, ws, npix, C, Cp);
            channel_sum_merge_kernel<<<(C + 255) / 256, 256, 0, st>>>(ws, out, (int)gx, C, accumulate);
            YDL_LAUNCH_CHECK();
            return 0;
        }
    }
    int nblk = (int)((npix + 127) / 128);
    if (nblk > CS_BLOCKS) nblk = CS_BLOCKS;
    const long long per = (npix + nblk - 1) / nblk;
    nblk = (int)((npix + per - 1) / per);
    dim3 grid(nblk, (C + 255) / 256);
    if (dtype == YDL_F32) channel_sum_part_kernel<float><<<grid, 256, 0, st>>>((const float*)x, ldx, ws, npix, C, per);
    else channel_sum_part_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, ldx, ws, npix, C, per);
    channel_sum_merge_kernel<<<(C + 255) / 256, 256, 0, st>>>(ws, out, nblk, C, accumulate);
    YDL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// soft-max over the P = K*K sampling points of each group: x, y are (npix, G*P) NHWC rows
// ------------------------------------------------------------------------------------------------------
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void group_softmax_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb,
                                                            T* __restrict__ o, int ldo, int accumulate, long long npix, int G, int P) {
    // FWD: a = logits, o = probabilities.   BWD: a = probabilities, b = d(prob), o = d(logits)
    const long long total = npix * G;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i % G);
        const long long pix = i / G;
        const T* ap = a + (size_t)pix * lda + g * P;
        T* op = o + (size_t)pix * ldo + g * P;
        if (!BWD) {
            float mx = -3.4e38f;
            for (int k = 0; k < P; ++k) mx = fmaxf(mx, ET<T>::ld(ap + k));
            float s = 0.f;
            for (int k = 0; k < P; ++k) s += __expf(ET<T>::ld(ap + k) - mx);
            const float r = 1.f / s;
            for (int k = 0; k < P; ++k) ET<T>::st(op + k, __expf(ET<T>::ld(ap + k) - mx) * r);
        } else {
            const T* bp = b + (size_t)pix * ldb + g * P;
            float dot = 0.f;
            for (int k = 0; k < P; ++k) dot = fmaf(ET<T>::ld(ap + k), ET<T>::ld(bp + k), dot);
            for (int k = 0; k < P; ++k) {
                float v = ET<T>::ld(ap + k) * (ET<T>::ld(bp + k) - dot);
                if (accumulate) v += ET<T>::ld(op + k);
                ET<T>::st(op + k, v);
            }
        }
    }
}
extern "C" int ydl_group_softmax_fwd(int dtype, const void* x, int ldx, void* y, int ldy, int64_t npix, int G, int P, void* stream) {
    YDL_CHECK((dtype == YDL_F32 || dtype == YDL_BF16) && x && y && npix > 0 && G > 0 && P > 0 && ldx >= G * P && ldy >= G * P, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int grid = stream_grid(npix * G);
    if (dtype == YDL_F32) group_softmax_kernel<float, false><<<grid, 256, 0, st>>>((const float*)x, ldx, nullptr, 0, (float*)y, ldy, 0, npix, G, P);
    else group_softmax_kernel<bf16_t, false><<<grid, 256, 0, st>>>((const bf16_t*)x, ldx, nullptr, 0, (bf16_t*)y, ldy, 0, npix, G, P);
    YDL_LAUNCH_CHECK();
    return 0;
}
extern "C" int ydl_group_softmax_bwd(int dtype, const void* y, int ldy, const void* dy, int lddy, void* dx, int lddx, int accumulate,
                                     int64_t npix, int G, int P, void* stream) {
    YDL_CHECK((dtype == YDL_F32 || dtype == YDL_BF16) && y && dy && dx && npix > 0 && G > 0 && P > 0, "bad arguments");
    YDL_CHECK(ldy >= G * P && lddy >= G * P && lddx >= G * P, "pixel strides must cover G*P");
    hipStream_t st = (hipStream_t)stream;
    const int grid = stream_grid(npix * G);
    if (dtype == YDL_F32) group_softmax_kernel<float, true><<<grid, 256, 0, st>>>((const float*)y, ldy, (const float*)dy, lddy, (float*)dx, lddx, accumulate, npix, G, P);
    else group_softmax_kernel<bf16_t, true><<<grid, 256, 0, st>>>((const bf16_t*)y, ldy, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, accumulate, npix, G, P);
    YDL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// dst[p][0:C] (op)= (T) src[p][0:C]   (src f32 with row stride lds, dst compute dtype with row stride ldd)
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cast_f32_kernel(const float* __restrict__ src, int lds_, T* __restrict__ dst, int ldd, long long npix,
                                                       int C, int accumulate) {
    const long long total = npix * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long pix = i / C;
        float v = src[(size_t)pix * lds_ + c];
        T* d = dst + (size_t)pix * ldd + c;
        if (accumulate) v += ET<T>::ld(d);
        ET<T>::st(d, v);
    }
}
extern "C" int ydl_cast_f32(int dtype, const float* src, int lds_, void* dst, int ldd, int64_t npix, int C, int accumulate, void* stream) {
    YDL_CHECK((dtype == YDL_F32 || dtype == YDL_BF16) && src && dst && npix > 0 && C > 0 && lds_ >= C && ldd >= C, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int grid = stream_grid(npix * C);
    if (dtype == YDL_F32) cast_f32_kernel<float><<<grid, 256, 0, st>>>(src, lds_, (float*)dst, ldd, npix, C, accumulate);
    else cast_f32_kernel<bf16_t><<<grid, 256, 0, st>>>(src, lds_, (bf16_t*)dst, ldd, npix, C, accumulate);
    YDL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------
// gradient transport helpers of yolo_dual_amd.parallel (the hand-rolled reduce-scatter + all-gather, bf16 wire format):
//   ydl_reduce_chunks: dst[i] (f32) = sum_r src[r][i], r = 0..nchunks-1 in that fixed order (src f32 or bf16)
//   ydl_cast_to_f32  : dst[i] (f32) (+)= src[i] (bf16 / f32)
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void reduce_chunks_kernel(const T* __restrict__ src, float* __restrict__ dst, long long n, int nchunks) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int r = 0; r < nchunks; ++r) s += ET<T>::ld(src + (size_t)r * n + i);
        dst[i] = s;
    }
}
extern "C" int ydl_reduce_chunks(int dtype, const void* src, float* dst, int64_t n, int nchunks, void* stream) {
    YDL_CHECK((dtype == YDL_F32 || dtype == YDL_BF16) && src && dst && n > 0 && nchunks > 0, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int grid = stream_grid(n);
    if (dtype == YDL_F32) reduce_chunks_kernel<float><<<grid, 256, 0, st>>>((const float*)src, dst, n, nchunks);
    else reduce_chunks_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)src, dst, n, nchunks);
    YDL_LAUNCH_CHECK();
    return 0;
}
template <typename T>
__global__ __launch_bounds__(256) void cast_to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, long long n, int accumulate) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = ET<T>::ld(src + i);
        dst[i] = accumulate ? dst[i] + v : v;
    }
}
extern "C" int ydl_cast_to_f32(int dtype, const void* src, float* dst, int64_t n, int accumulate, void* stream) {
    YDL_CHECK((dtype == YDL_F32 || dtype == YDL_BF16) && src && dst && n > 0, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int grid = stream_grid(n);
    if (dtype == YDL_F32) cast_to_f32_kernel<float><<<grid, 256, 0, st>>>((const float*)src, dst, n, accumulate);
    else cast_to_f32_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)src, dst, n, accumulate);
    YDL_LAUNCH_CHECK();
    return 0;
}
