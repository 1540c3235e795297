// Weight re-layout (f32 KRSC master -> compute copies), wgrad un-padding, and the fused SGD(nesterov)+EMA step
// over flat arenas (utils/torch_utils.py:318-346 smart_optimizer groups, :404-428 ModelEMA.update).
#include "common.h"

// w[co][t][ci_p] = master[co][t][ci] (zero pad);  wt[ci][t][co_p] = master[co][t][ci]
template <typename T>
__global__ __launch_bounds__(256) void weight_prep_kernel(const float* __restrict__ master, T* __restrict__ w, T* __restrict__ wt,
                                                          int Cout, int kk, int Cin, int Cin_p, int Cout_p) {
    const long long n1 = (long long)Cout * kk * Cin_p;
    const long long n2 = (long long)Cin * kk * Cout_p;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += (long long)gridDim.x * blockDim.x) {
        if (i < n1) {
            if (w == nullptr) continue;
            int ci = (int)(i % Cin_p);
            long long r = i / Cin_p;           // co*kk + t
            float v = ci < Cin ? master[r * Cin + ci] : 0.f;
            ET<T>::st(w + i, v);
        } else {
            if (wt == nullptr) continue;
            long long j = i - n1;
            int co = (int)(j % Cout_p);
            long long r = j / Cout_p;          // ci*kk + t
            int t = (int)(r % kk);
            int ci = (int)(r / kk);
            float v = co < Cout ? master[((long long)co * kk + t) * Cin + ci] : 0.f;
            ET<T>::st(wt + j, v);
        }
    }
}

extern "C" int ydl_weight_prep(int dtype, const float* master, void* w, void* wt, int Cout, int kk, int Cin, void* stream) {
    YDL_CHECK(master && (w || wt) && Cout > 0 && kk > 0 && Cin > 0, "bad arguments");
    int Cin_p = round_up(Cin, 8), Cout_p = round_up(Cout, 8);
    long long total = (long long)Cout * kk * Cin_p + (long long)Cin * kk * Cout_p;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YDL_F32) weight_prep_kernel<float><<<grid, 256, 0, st>>>(master, (float*)w, (float*)wt, Cout, kk, Cin, Cin_p, Cout_p);
    else weight_prep_kernel<bf16_t><<<grid, 256, 0, st>>>(master, (bf16_t*)w, (bf16_t*)wt, Cout, kk, Cin, Cin_p, Cout_p);
    YDL_LAUNCH_CHECK();
    return 0;
}

// all layers of a model in ONE launch: blockIdx.y = layer, descriptors live in device memory
// desc[l] = {master*, w*, wt*, Cout, kk, Cin, 0, 0} as 8 x int64.
// Per tap the [Cout][Cin] matrix is walked in 32x32 tiles through LDS: the master rows are read coalesced once, `w` is
// written in the same order and `wt` transposed, both coalesced (the former element-wise version read the master with a
// k*k*Cin stride for `wt`: 549 MB fetched per step for 68 MB of weights in the PMC profile).
template <typename T>
__global__ __launch_bounds__(256) void weight_prep_batched_kernel(const long long* __restrict__ desc, int nlayers) {
    // tiles of ALL layers in one flat index space (one grid row per layer gave the 4.7 M-element layers 256 CTAs like the 8 K-element
    // ones: the launch took as long as its largest layer, 58 us for 68 MB in / 68 MB out).  Every CTA derives the per-layer tile
    // prefix itself: layer = thread, inclusive scan through LDS.
    __shared__ int s_end[256];
    __shared__ float tile[32][33];
    const int tid = threadIdx.x;
    {
        int nt = 0;
        if (tid < nlayers) {
            const long long* d = desc + (size_t)tid * 8;
            const int Cout = (int)d[3], kk = (int)d[4], Cin = (int)d[5];
            nt = ((Cout + 7) / 8 * 8 + 31) / 32 * (((Cin + 7) / 8 * 8 + 31) / 32) * kk;
        }
        s_end[tid] = nt;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const int v = tid >= o ? s_end[tid - o] : 0;
            __syncthreads();
            s_end[tid] += v;
            __syncthreads();
        }
    }
    const int total = s_end[nlayers - 1];
    const int tx = tid & 31, ty = tid >> 5;
    int l = 0;
    for (int gt = blockIdx.x; gt < total; gt += gridDim.x) {
        while (gt >= s_end[l]) ++l;                       // gt only grows
        const long long* d = desc + (size_t)l * 8;
        const float* master = (const float*)d[0];
        T* w = (T*)d[1];
        T* wt = (T*)d[2];
        const int Cout = (int)d[3], kk = (int)d[4], Cin = (int)d[5];
        const int Cin_p = (Cin + 7) / 8 * 8, Cout_p = (Cout + 7) / 8 * 8;
        const int tci = (Cin_p + 31) / 32;
        const int tl = gt - (l ? s_end[l - 1] : 0);
        const int cit = tl % tci;
        const int r2 = tl / tci;
        const int t = r2 % kk;
        const int cot = r2 / kk;
        const int co0 = cot * 32, ci0 = cit * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + ty + 8 * i, ci = ci0 + tx;
            float v = (co < Cout && ci < Cin) ? master[((size_t)co * kk + t) * Cin + ci] : 0.f;
            tile[ty + 8 * i][tx] = v;
            if (co < Cout && ci < Cin_p) ET<T>::st(w + ((size_t)co * kk + t) * Cin_p + ci, v);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ci = ci0 + ty + 8 * i, co = co0 + tx;
            if (ci < Cin && co < Cout_p) ET<T>::st(wt + ((size_t)ci * kk + t) * Cout_p + co, tile[tx][ty + 8 * i]);
        }
        __syncthreads();
    }
}

extern "C" int ydl_weight_prep_batched(int dtype, const int64_t* desc_dev, int nlayers, void* stream) {
    YDL_CHECK(desc_dev && nlayers > 0, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    for (int l0 = 0; l0 < nlayers; l0 += 256) {            // 256 layers per launch (one scan lane each)
        const int nl = nlayers - l0 < 256 ? nlayers - l0 : 256;
        const long long* d = (const long long*)desc_dev + (size_t)l0 * 8;
        if (dtype == YDL_F32) weight_prep_batched_kernel<float><<<2048, 256, 0, st>>>(d, nl);
        else weight_prep_batched_kernel<bf16_t><<<2048, 256, 0, st>>>(d, nl);
    }
    YDL_LAUNCH_CHECK();
    return 0;
}

__global__ void wgrad_unpad_kernel(const float* __restrict__ dw, float* __restrict__ grad, long long rows, int Cin, int Cin_p, int accumulate) {
    const long long total = rows * Cin;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long r = i / Cin;
        int c = (int)(i - r * Cin);
        float v = dw[r * Cin_p + c];
        grad[i] = accumulate ? grad[i] + v : v;
    }
}
extern "C" int ydl_wgrad_unpad(const float* dw, float* grad, int Cout, int kk, int Cin, int accumulate, void* stream) {
    YDL_CHECK(dw && grad, "null pointer");
    long long rows = (long long)Cout * kk;
    long long total = rows * Cin;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    wgrad_unpad_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(dw, grad, rows, Cin, round_up(Cin, 8), accumulate);
    YDL_LAUNCH_CHECK();
    return 0;
}

// space-to-depth form of a stem weight (see ydl_nchw_to_s2d): master [Cout][k][k][C] (KRSC) ->
//   w2[co][(r2*k2 + c2)][(dy*s+dx)*C + c] = master[co][s*r2+dy][s*c2+dx][c],   k2 = k/s, row padded to Cs_p = round_up(s*s*C, 8)
template <typename T>
__global__ void weight_prep_s2d_kernel(const float* __restrict__ master, T* __restrict__ w2, int Cout, int k, int s, int C) {
    const int k2 = k / s, Cs = C * s * s, Cs_p = (Cs + 7) / 8 * 8;
    const long long total = (long long)Cout * k2 * k2 * Cs_p;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % Cs_p);
        long long r = i / Cs_p;
        const int c2 = (int)(r % k2); r /= k2;
        const int r2 = (int)(r % k2);
        const int co = (int)(r / k2);
        float v = 0.f;
        if (ch < Cs) {
            const int c = ch % C, dd = ch / C, dx = dd % s, dy = dd / s;
            v = master[(((size_t)co * k + (s * r2 + dy)) * k + (s * c2 + dx)) * C + c];
        }
        ET<T>::st(w2 + i, v);
    }
}
extern "C" int ydl_weight_prep_s2d(int dtype, const float* master, void* w2, int Cout, int k, int s, int C, void* stream) {
    YDL_CHECK(master && w2 && Cout > 0 && C > 0 && s >= 1 && k % s == 0, "k must be a multiple of s");
    const int k2 = k / s;
    long long total = (long long)Cout * k2 * k2 * round_up(C * s * s, 8);
    int grid = (int)((total + 255) / 256);
    if (grid > 1024) grid = 1024;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == YDL_F32) weight_prep_s2d_kernel<float><<<grid, 256, 0, st>>>(master, (float*)w2, Cout, k, s, C);
    else weight_prep_s2d_kernel<bf16_t><<<grid, 256, 0, st>>>(master, (bf16_t*)w2, Cout, k, s, C);
    YDL_LAUNCH_CHECK();
    return 0;
}

// the inverse index map for the weight gradient: grad[co][s*r2+dy][s*c2+dx][c] (+)= dw2[co][r2*k2+c2][(dy*s+dx)*C + c]
__global__ void wgrad_unpack_s2d_kernel(const float* __restrict__ dw2, float* __restrict__ grad, int Cout, int k, int s, int C,
                                        int accumulate) {
    const int k2 = k / s, Cs = C * s * s, Cs_p = (Cs + 7) / 8 * 8;
    const long long total = (long long)Cout * k * k * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int kw = (int)(r % k); r /= k;
        const int kh = (int)(r % k);
        const int co = (int)(r / k);
        const int r2 = kh / s, dy = kh % s, c2 = kw / s, dx = kw % s;
        const float v = dw2[(((size_t)co * k2 + r2) * k2 + c2) * Cs_p + (dy * s + dx) * C + c];
        grad[i] = accumulate ? grad[i] + v : v;
    }
}
extern "C" int ydl_wgrad_unpack_s2d(const float* dw2, float* grad, int Cout, int k, int s, int C, int accumulate, void* stream) {
    YDL_CHECK(dw2 && grad && Cout > 0 && C > 0 && s >= 1 && k % s == 0, "k must be a multiple of s");
    long long total = (long long)Cout * k * k * C;
    int grid = (int)((total + 255) / 256);
    if (grid > 1024) grid = 1024;
    wgrad_unpack_s2d_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(dw2, grad, Cout, k, s, C, accumulate);
    YDL_LAUNCH_CHECK();
    return 0;
}

// One pass over [params | buffers]: SGD-nesterov on params (two lr/decay groups), EMA on everything.
//   g = grad*grad_scale + wd*p ; buf = first ? g : mom*buf + g ; p -= lr*(g + mom*buf) ; ema = d*ema + (1-d)*p
__global__ __launch_bounds__(256) void sgd_ema_kernel(float* __restrict__ params, const float* __restrict__ grads,
                                                      float* __restrict__ mombuf, float* __restrict__ ema,
                                                      long long n_decay, long long n_params, long long n_total,
                                                      float lr0, float lr1, float mom, float wd, float gscale, int first, float d) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_total; i += (long long)gridDim.x * blockDim.x) {
        float p = params[i];
        if (i < n_params) {
            bool dec = i < n_decay;
            float g = grads[i] * gscale;
            if (dec && wd != 0.f) g = g + wd * p;
            float b = first ? g : mom * mombuf[i] + g;
            mombuf[i] = b;
            float upd = g + mom * b;
            p = p - (dec ? lr0 : lr1) * upd;
            params[i] = p;
        }
        if (d >= 0.f && ema != nullptr) {
            float e = ema[i];
            e = e * d;
            e = e + (1.f - d) * p;
            ema[i] = e;
        }
    }
}

extern "C" int ydl_sgd_ema_step(float* params, const float* grads, float* momentum, float* ema,
                                int64_t n_decay, int64_t n_params, int64_t n_total,
                                float lr_decay_group, float lr_nodecay_group, float mom, float weight_decay, float grad_scale,
                                int first_step, float ema_decay, void* stream) {
    YDL_CHECK(params && grads && momentum, "null pointer");
    YDL_CHECK(0 <= n_decay && n_decay <= n_params && n_params <= n_total, "bad arena partition");
    int grid = (int)((n_total + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    sgd_ema_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(params, grads, momentum, ema, n_decay, n_params, n_total,
                                                           lr_decay_group, lr_nodecay_group, mom, weight_decay, grad_scale,
                                                           first_step, ema_decay);
    YDL_LAUNCH_CHECK();
    return 0;
}


// Same update with the hyper-parameters read from DEVICE memory, so that the launch can live inside a captured HIP graph
// while lr / EMA decay change every step.  hyper = {lr_weights, lr_bn, lr_bias, momentum, weight_decay, grad_scale, ema_d}
__global__ __launch_bounds__(256) void sgd_ema_dev_kernel(float* __restrict__ params, const float* __restrict__ grads,
                                                          float* __restrict__ mombuf, float* __restrict__ ema,
                                                          long long n_decay, long long n_params, long long n_total,
                                                          const float* __restrict__ hyper, int lr_idx, int use_wd, int first,
                                                          int use_ema) {
    const float lr = hyper[lr_idx], mom = hyper[3], wd = use_wd ? hyper[4] : 0.f, gscale = hyper[5], d = hyper[6];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_total; i += (long long)gridDim.x * blockDim.x) {
        float p = params[i];
        if (i < n_params) {
            bool dec = i < n_decay;
            float g = grads[i] * gscale;
            if (dec && wd != 0.f) g = g + wd * p;
            float b = first ? g : mom * mombuf[i] + g;
            mombuf[i] = b;
            float upd = g + mom * b;
            p = p - lr * upd;
            params[i] = p;
        }
        if (use_ema && ema != nullptr) {
            float e = ema[i];
            e = e * d;
            e = e + (1.f - d) * p;
            ema[i] = e;
        }
    }
}

// All runs of one optimizer step in ONE launch: grid row y handles run y = {offset, n_decay, n_params, n_total, lr_index,
// flags (bit 0 weight decay, bit 1 first step)} of the device table (the per-run launches of the dead / live / BN / bias / buffer
// ranges cost ~7 us each, nine of them per step on BASELINE config 2).  Same arithmetic as sgd_ema_dev_kernel.
__device__ __forceinline__ void sgd_ema_elem(float& p, float g, float& b, float& e, bool is_param, bool dec, bool first, bool use_ema,
                                             float lr, float mom, float wd, float gscale, float d) {
    if (is_param) {
        g = g * gscale;
        if (dec && wd != 0.f) g = g + wd * p;
        b = first ? g : mom * b + g;
        const float upd = g + mom * b;
        p = p - lr * upd;
    }
    if (use_ema) {
        e = e * d;
        e = e + (1.f - d) * p;
    }
}
__global__ __launch_bounds__(256) void sgd_ema_multi_kernel(float* __restrict__ params, const float* __restrict__ grads,
                                                            float* __restrict__ mombuf, float* __restrict__ ema,
                                                            const long long* __restrict__ runs, const float* __restrict__ hyper,
                                                            int use_ema) {
    const long long* r = runs + (size_t)blockIdx.y * 6;
    const long long off = r[0], n_decay = r[1], n_params = r[2], n_total = r[3];
    const int lr_idx = (int)r[4], flags = (int)r[5];
    const bool first = (flags >> 1) & 1;
    const float lr = hyper[lr_idx], mom = hyper[3], wd = (flags & 1) ? hyper[4] : 0.f, gscale = hyper[5], d = hyper[6];
    const bool ue = use_ema && ema != nullptr;
    // 16 bytes per lane and stream (seven streams: p rw, g r, momentum rw, ema rw): the run is walked in float4 groups aligned to the
    // ARENA (so every access is 16-byte aligned whatever the run's offset); the groups that straddle the run's ends, and runs whose
    // decay / parameter boundary is not the whole run, take the element form.  Same arithmetic per element either way.
    const long long a0 = off & ~3ll, end = off + n_total;
    const bool uniform = (n_params == n_total || n_params == 0) && (n_decay == n_params || n_decay == 0);
    for (long long q = a0 + 4 * ((long long)blockIdx.x * blockDim.x + threadIdx.x); q < end; q += 4ll * gridDim.x * blockDim.x) {
        if (uniform && q >= off && q + 4 <= end) {
            const bool isp = n_params != 0, dec = n_decay != 0;
            float4 p4 = *(const float4*)(params + q), g4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = g4, e4 = g4;
            if (isp) { g4 = *(const float4*)(grads + q); if (!first) b4 = *(const float4*)(mombuf + q); }
            if (ue) e4 = *(const float4*)(ema + q);
            sgd_ema_elem(p4.x, g4.x, b4.x, e4.x, isp, dec, first, ue, lr, mom, wd, gscale, d);
            sgd_ema_elem(p4.y, g4.y, b4.y, e4.y, isp, dec, first, ue, lr, mom, wd, gscale, d);
            sgd_ema_elem(p4.z, g4.z, b4.z, e4.z, isp, dec, first, ue, lr, mom, wd, gscale, d);
            sgd_ema_elem(p4.w, g4.w, b4.w, e4.w, isp, dec, first, ue, lr, mom, wd, gscale, d);
            if (isp) { *(float4*)(params + q) = p4; *(float4*)(mombuf + q) = b4; }
            if (ue) *(float4*)(ema + q) = e4;
        } else {
            for (int k = 0; k < 4; ++k) {
                const long long i = q + k - off;
                if (i < 0 || i >= n_total) continue;
                const bool isp = i < n_params;
                float p = params[off + i], g = 0.f, b = 0.f, e = 0.f;
                if (isp) { g = grads[off + i]; if (!first) b = mombuf[off + i]; }
                if (ue) e = ema[off + i];
                sgd_ema_elem(p, g, b, e, isp, i < n_decay, first, ue, lr, mom, wd, gscale, d);
                if (isp) { params[off + i] = p; mombuf[off + i] = b; }
                if (ue) ema[off + i] = e;
            }
        }
    }
}

extern "C" int ydl_sgd_ema_step_multi(float* params, const float* grads, float* momentum, float* ema, const int64_t* runs_dev,
                                      int nruns, int64_t max_run, const float* hyper_dev, int use_ema, void* stream) {
    YDL_CHECK(params && grads && momentum && runs_dev && hyper_dev && nruns > 0 && nruns <= 65535 && max_run >= 0, "bad arguments");
    long long gx = (max_run + 4 + 1023) / 1024;           // four elements per thread (+ the alignment slack of the first group)
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    sgd_ema_multi_kernel<<<dim3((unsigned)gx, (unsigned)nruns), 256, 0, (hipStream_t)stream>>>(params, grads, momentum, ema,
                                                                                             (const long long*)runs_dev, hyper_dev, use_ema);
    YDL_LAUNCH_CHECK();
    return 0;
}

extern "C" int ydl_sgd_ema_step_dev(float* params, const float* grads, float* momentum, float* ema,
                                    int64_t n_decay, int64_t n_params, int64_t n_total, const float* hyper_dev,
                                    int lr_index, int use_weight_decay, int first_step, int use_ema, void* stream) {
    YDL_CHECK(params && grads && momentum && hyper_dev, "null pointer");
    YDL_CHECK(0 <= n_decay && n_decay <= n_params && n_params <= n_total && lr_index >= 0 && lr_index <= 2, "bad arguments");
    int grid = (int)((n_total + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    sgd_ema_dev_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(params, grads, momentum, ema, n_decay, n_params, n_total,
                                                               hyper_dev, lr_index, use_weight_decay, first_step, use_ema);
    YDL_LAUNCH_CHECK();
    return 0;
}
