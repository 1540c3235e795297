// Shared device/host helpers for libydl_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "ydl.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef unsigned short bf16_t;  // storage type of a bf16 element

#define YDL_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(unsigned short, b);
}

// element traits: V = elements per 16-byte chunk
template <typename T> struct ET;
template <> struct ET<float> {
    static constexpr int V = 4;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ET<bf16_t> {
    static constexpr int V = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 16-byte chunk <-> floats
template <typename T> __device__ __forceinline__ void unpack16(const uint4& u, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* f);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* f) {
    uint4 u;
    u.x = (uint32_t)f2bf(f[0]) | ((uint32_t)f2bf(f[1]) << 16);
    u.y = (uint32_t)f2bf(f[2]) | ((uint32_t)f2bf(f[3]) << 16);
    u.z = (uint32_t)f2bf(f[4]) | ((uint32_t)f2bf(f[5]) << 16);
    u.w = (uint32_t)f2bf(f[6]) | ((uint32_t)f2bf(f[7]) << 16);
    return u;
}

// v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the IEEE division / range-reduced exp sequences (about 25 VALU instructions per
// element, which made the streaming BN kernels VALU-bound).  exp2 overflow gives rcp(inf) = 0, underflow gives rcp(1) = 1.
__device__ __forceinline__ float sigmoid_f(float z) {
#ifdef YDL_FAKE_SIGMOID      // timing experiment only (what the two transcendentals cost the streaming kernels): wrong results
    return 0.5f + 0.125f * z;
#else
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
#endif
}
__device__ __forceinline__ float silu_f(float z) { return z * sigmoid_f(z); }

// BatchNorm backward, per 16-byte bf16 chunk: dz = dout * act'(y*scale + shift), xhat = (y - mean) * invstd (bn.hip: dz_xhat_q;
// also used by the convolution epilogues that fuse the reduce pass: ydl_conv_dgrad_bnred)
__device__ __forceinline__ void bn_dz_xhat_bf16x8(const uint4& yq, const uint4& dq, const float* sc, const float* sf, const float* mu,
                                                  const float* is, bool silu, float* dz, float* xh) {
    float yv[8], dv[8];
    unpack16<bf16_t>(yq, yv);
    unpack16<bf16_t>(dq, dv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float z = yv[e] * sc[e] + sf[e];
        float d = dv[e];
        if (silu) {
            const float sg = sigmoid_f(z);
            d *= sg * (1.f + z * (1.f - sg));
        }
        dz[e] = d;
        xh[e] = (yv[e] - mu[e]) * is[e];
    }
}

// Sum over the 64 lanes with DPP row shifts + row broadcasts (six v_add_f32_dpp, no LDS crossbar): the total is valid in LANE 63 ONLY.
// The ds_bpermute butterfly of wave_sum below occupies the LDS for every step: the loss kernels reduce 39 values per wave with it and
// spent half their time there (seg_loss_rep_fwd: 25 % LDS-active, 58 us; tools/sq_summary.py).
__device__ __forceinline__ float wave_total63(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));     // row_shr:1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, false));     // row_shr:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xe, false));     // row_shr:4 (banks 1..3)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xc, false));     // row_shr:8 (banks 2..3)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));     // row_bcast:15 (rows 1, 3)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));     // row_bcast:31 (rows 2, 3)
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- host-side error plumbing -----------------------------------------------------------------------
void ydl_set_error(const std::string& s);
#define YDL_CHECK(cond, msg)                                                                 \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            ydl_set_error(std::string(__func__) + ": " + (msg) + " [" #cond "]");           \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)
#define YDL_LAUNCH_CHECK()                                                                   \
    do {                                                                                     \
        hipError_t e_ = hipGetLastError();                                                   \
        if (e_ != hipSuccess) {                                                              \
            ydl_set_error(std::string(__func__) + ": launch failed: " + hipGetErrorString(e_)); \
            return 2;                                                                        \
        }                                                                                    \
    } while (0)

// ---- per-device state --------------------------------------------------------------------------------
// One process may drive several GPUs (the reference's nn.DataParallel does): launch attributes and device properties
// are kept per HIP device id.  ydl_dev_once(tag) returns true exactly once per (tag, current device) — the caller then
// sets its kernel attributes for THAT device; ydl_device_cus() is the CU count of the current device.
#define YDL_MAX_DEVICES 64
bool ydl_dev_once(void* tag_storage);      // tag_storage: a static std::atomic<uint64_t>-sized word owned by the call site
int ydl_device_cus();
void ydl_note_kernel(int family, const char* name);   // diagnostics: last kernel instantiation launched per entry family
#define YDL_SET_MAX_LDS(kernel, bytes)                                                                              \
    do {                                                                                                            \
        static unsigned long long once_ = 0;                                                                        \
        if (ydl_dev_once(&once_)) {                                                                                 \
            hipError_t e_ = hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            if (e_ != hipSuccess) {                                                                                 \
                ydl_set_error(std::string(__func__) + ": hipFuncSetAttribute failed: " + hipGetErrorString(e_));   \
                return 2;                                                                                           \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
static inline int esize(int dtype) { return dtype == YDL_F32 ? 4 : 2; }
