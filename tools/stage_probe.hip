// Probe (dev tool): how fast can ONE CU pull L2-resident bytes, by path?  One 512-thread workgroup per CU (or two 256-thread ones), every
// XCD's workgroups sweep the same 2 MB window (L2-resident after the first pass), 64 KB per step:
//   mode 0  buffer_load_dwordx4 ... lds (LDS-DMA), two 64 KB stages in flight
//   mode 1  buffer_load_dwordx4 into registers (xor-reduced), two sets of 8 in flight
//   mode 2  registers + ds_write_b128 (register staging into the same LDS image)
//   mode 3  half the bytes by LDS-DMA, half into registers (do the two paths add up?)
//   mode 4  mode 0 with an s_barrier per step (what a ring kernel does)
// build: hipcc --offload-arch=gfx950 -O3 tools/stage_probe.hip -o tools/bin/stage_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(u32x4 rs, unsigned lds, unsigned off) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds), "v"(off), "s"(rs) : "memory");
}
__device__ __forceinline__ uint4 ld16(u32x4 rs, unsigned off) {
    uint4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(off), "s"(rs) : "memory");
    return v;
}

template <int MODE, int NT, int CH = 65536, int ROWS = 0>
__global__ __launch_bounds__(NT) void probe(const unsigned char* src, unsigned window, int steps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PER = CH / (NT * 16);                      // 16-byte pieces per thread and 64 KB step
    const unsigned long long pa = (unsigned long long)src;
    const u32x4 rs = u32x4{(unsigned)pa, (unsigned)(pa >> 32) & 0xffffu, window, 0x00020000u};
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned nchunk = window / CH;
    unsigned acc = 0;
    uint4 r[2][PER];
    auto chunk_off = [&](int s) { return (((unsigned)s * 37u + blockIdx.x * 5u) % nchunk) * (unsigned)CH; };
    auto issue = [&](int s) {
        const unsigned base = chunk_off(s);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            unsigned off = base + (unsigned)((i * NT + t) * 16);
            if (ROWS) { const unsigned e = (unsigned)(i * NT + t); off = (base + (e >> 3) * 4224u + (e & 7u) * 16u) % window; }
            const unsigned l = lds0 + (unsigned)(s & 1) * (unsigned)CH + (unsigned)((i * NT + wave * 64) * 16);
            if (MODE == 0 || MODE == 4) dma16(rs, l, off);
            else if (MODE == 3) { if (i & 1) dma16(rs, l, off); else r[s & 1][i] = ld16(rs, off); }
            else r[s & 1][i] = ld16(rs, off);
        }
    };
    auto consume = [&](int s) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (MODE == 1 || (MODE == 3 && !(i & 1))) acc ^= r[s & 1][i].x ^ r[s & 1][i].y ^ r[s & 1][i].z ^ r[s & 1][i].w;
            if (MODE == 2) *(uint4*)(smem + (s & 1) * CH + (i * NT + t) * 16) = r[s & 1][i];
        }
    };
    issue(0);
    for (int s = 0; s < steps; ++s) {
        issue(s + 1);
        if (MODE == 0 || MODE == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        else if (MODE == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        if (MODE == 4) __builtin_amdgcn_s_barrier();
        consume(s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    consume(steps);
    if (MODE == 0 || MODE == 2 || MODE == 3 || MODE == 4) { __syncthreads(); acc ^= *(unsigned*)(smem + ((t * 16 + lane) & 0x1fff0)); }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int NT, int CH = 65536, int ROWS = 0>
static void run(const char* name, const unsigned char* d, unsigned window, int ctas, int steps, unsigned* sink) {
    hipFuncSetAttribute((const void*)probe<MODE, NT, CH, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CH);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, NT, CH, ROWS><<<ctas, NT, 2 * CH>>>(d, window, steps, sink);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        probe<MODE, NT, CH, ROWS><<<ctas, NT, 2 * CH>>>(d, window, steps, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)ctas * (steps + 1) * (double)CH;
    printf("%-44s %4d CTAs x %3d thr  %8.3f ms  %7.1f GB/s per CU  %6.2f TB/s chip\n", name, ctas, NT, best, bytes / best / 1e6 / 256.0,
           bytes / best / 1e9);
}


// mode 5: the weight-gradient kernels' shape — 256-thread workgroups, two per CU, a ring of S stages of 16 KB, one counted wait + barrier
// per stage; SHARE consecutive workgroups stream the SAME bytes (the column tiles of one pixel chunk), each group its own contiguous range
template <int S, int SHARE>
__global__ __launch_bounds__(256, 2) void stream_probe(const unsigned char* src, unsigned long long bytes, int steps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ST = 16384, PER = ST / (256 * 16);
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int L = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);          // consecutive L on one XCD
    const unsigned group = (unsigned)L / SHARE, ngroups = (gridDim.x + SHARE - 1) / SHARE;
    const unsigned long long span = (bytes / ngroups) & ~0xFFFFull;
    const unsigned char* base = src + (unsigned long long)group * span;
    const unsigned long long pa = (unsigned long long)base;
    const u32x4 rs = u32x4{(unsigned)pa, (unsigned)(pa >> 32) & 0xffffu, (unsigned)span, 0x00020000u};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    auto issue = [&](int s) {
        const unsigned off0 = (unsigned)(((unsigned long long)s * ST) % span);
#pragma unroll
        for (int i = 0; i < PER; ++i)
            dma16(rs, lds0 + (unsigned)(s % S) * ST + (unsigned)((i * 256 + wave * 64) * 16), off0 + (unsigned)((i * 256 + t) * 16));
    };
#pragma unroll
    for (int u = 0; u < S - 1; ++u) issue(u);
    for (int s = 0; s < steps; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PER * (S - 2)) : "memory");
        issue(s + S - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (*(unsigned*)(smem + t * 16) == 0x12345678u) sink[0] = 1;
}
template <int S, int SHARE>
static void run_stream(const char* name, const unsigned char* d, unsigned long long bytes, int steps, unsigned* sink) {
    hipFuncSetAttribute((const void*)stream_probe<S, SHARE>, hipFuncAttributeMaxDynamicSharedMemorySize, S * 16384);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        stream_probe<S, SHARE><<<512, 256, S * 16384>>>(d, bytes, steps, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double moved = 512.0 * (steps + S - 1) * 16384.0;
    printf("%-58s %8.3f ms  %6.1f GB/s per CU into LDS  %6.2f TB/s distinct  %.2f us per stage\n", name, best, moved / best / 1e6 / 256.0,
           moved / SHARE / best / 1e9, best * 1e3 / steps);
}

int main(int argc, char** argv) {
    const unsigned window = 2u << 20;
    unsigned char* d; unsigned* sink;
    hipMalloc(&d, window); hipMalloc(&sink, 64);
    hipMemset(d, 1, window);
    const int steps = argc > 1 ? atoi(argv[1]) : 400;
    run<0, 512>("LDS-DMA, 1 x 512", d, window, 256, steps, sink);
    run<4, 512>("LDS-DMA + s_barrier per step, 1 x 512", d, window, 256, steps, sink);
    run<1, 512>("registers, 1 x 512", d, window, 256, steps, sink);
    run<2, 512>("registers + ds_write_b128, 1 x 512", d, window, 256, steps, sink);
    run<3, 512>("half LDS-DMA, half registers, 1 x 512", d, window, 256, steps, sink);
    run<0, 256>("LDS-DMA, 1 x 256", d, window, 256, steps, sink);
    run<1, 256>("registers, 1 x 256", d, window, 256, steps, sink);
    run<0, 1024>("LDS-DMA, 1 x 1024", d, window, 256, steps, sink);
    run<1, 1024>("registers, 1 x 1024", d, window, 256, steps, sink);
    run<0, 128>("LDS-DMA, 1 x 128 (2 waves)", d, window, 256, steps, sink);
    run<0, 512, 32768>("LDS-DMA, 32 KB steps (32-64 KB in flight)", d, window, 256, steps * 2, sink);
    run<0, 512, 16384>("LDS-DMA, 16 KB steps (16-32 KB in flight)", d, window, 256, steps * 4, sink);
    run<0, 512, 8192>("LDS-DMA, 8 KB steps (8-16 KB in flight)", d, window, 256, steps * 8, sink);
    run<0, 256, 32768>("LDS-DMA, 256 thr, 32 KB steps", d, window, 256, steps * 2, sink);
    run<0, 256, 32768>("LDS-DMA, 2 x 256 thr per CU, 32 KB steps", d, window, 512, steps * 2, sink);
    run<0, 512, 65536, 1>("LDS-DMA, 8 rows x 128 B per instruction", d, window, 256, steps, sink);
    run<4, 512, 32768>("LDS-DMA + barrier, 32 KB steps", d, window, 256, steps * 2, sink);
    {
        unsigned char* big;
        const unsigned long long nb = 1ull << 30;
        hipMalloc(&big, nb); hipMemset(big, 1, nb);
        const int st = 600;
        run_stream<4, 1>("ring 4 x 16 KB, every workgroup its own stream, 1 GB", big, nb, st, sink);
        run_stream<4, 9>("ring 4 x 16 KB, 9 workgroups share a stream, 1 GB", big, nb, st, sink);
        run_stream<8, 9>("ring 8 x 16 KB (one workgroup per CU), 9 share, 1 GB", big, nb, st, sink);
        run_stream<4, 1>("ring 4 x 16 KB, own streams, 64 MB (Infinity Cache)", big, 64ull << 20, st, sink);
        run_stream<4, 9>("ring 4 x 16 KB, 9 share, 64 MB (Infinity Cache)", big, 64ull << 20, st, sink);
        run_stream<4, 9>("ring 4 x 16 KB, 9 share, 16 MB (L2: 2 MB per XCD)", big, 16ull << 20, st, sink);
        run_stream<2, 9>("ring 2 x 16 KB, 9 share, 64 MB", big, 64ull << 20, st, sink);
        run_stream<3, 9>("ring 3 x 16 KB, 9 share, 64 MB", big, 64ull << 20, st, sink);
        run_stream<5, 9>("ring 5 x 16 KB, 9 share, 64 MB", big, 64ull << 20, st, sink);
    }
    return 0;
}
