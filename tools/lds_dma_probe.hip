// Probe (dev tool): does an out-of-range `buffer_load_dwordx4 ... lds` lane write zeros into LDS or leave it untouched?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k1(const uint4* src, uint4* dst, unsigned nbytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    unsigned off = threadIdx.x * 16;
    if (threadIdx.x & 1) off = 0xFFFFFFFFu;
    if ((threadIdx.x & 3) == 2) off = nbytes + 64;      // beyond num_records by a small amount
    *(uint4*)(smem + threadIdx.x * 16) = make_uint4(7, 7, 7, 7);   // poison
    __syncthreads();
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    dst[threadIdx.x] = *(uint4*)(smem + threadIdx.x * 16);
}
int main() {
    const int n = 64;
    std::vector<uint4> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = make_uint4(100 + i, 100 + i, 100 + i, 100 + i);
    uint4 *d, *r;
    hipMalloc(&d, n * 16); hipMalloc(&r, n * 16);
    hipMemcpy(d, h.data(), n * 16, hipMemcpyHostToDevice);
    k1<<<1, 64, 4096>>>(d, r, n * 16);
    hipMemcpy(o.data(), r, n * 16, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("lane %d: %u %u %u %u\n", i, o[i].x, o[i].y, o[i].z, o[i].w);
    int zeros = 0, poison = 0, other = 0;
    for (int i = 0; i < n; ++i) if (i & 1 || (i & 3) == 2) { if (o[i].x == 0 && o[i].w == 0) ++zeros; else if (o[i].x == 7) ++poison; else ++other; }
    printf("out-of-range lanes: %d zero-filled, %d untouched (poison), %d other\n", zeros, poison, other);
    return 0;
}
