// GPU half of the reference's per-sample input preparation: JSONSegmentDataset._resize_and_pad and the
// format conversion of __getitem__ (unet-lite/yolo5-seg/seg_diceloss_yolov5.py:309-349).  Bit-exact with Pillow's 8-bit
// resampling: the 22-bit fixed-point coefficient tables and the nearest-neighbour index tables are built on the host in double
// (yolo_dual_amd/data.py), the kernels only do the integer arithmetic.  HBM-bound byte work: one thread per output sample.
#include "common.h"

#define PIL_PRECISION_BITS 22

__device__ __forceinline__ unsigned char pil_clip8(int ss) {
    int v = ss >> PIL_PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass of ImagingResample: src [h][w][3] -> tmp [h][new_w][3]
__global__ __launch_bounds__(256) void letterbox_h_kernel(const unsigned char* __restrict__ src, int h, int w,
                                                          unsigned char* __restrict__ tmp, int new_w,
                                                          const int* __restrict__ xb, const int* __restrict__ xk, int ksize) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)h * new_w) return;
    const int y = (int)(i / new_w), xx = (int)(i - (long long)y * new_w);
    const int x0 = xb[2 * xx], n = xb[2 * xx + 1];
    const int* k = xk + (size_t)xx * ksize;
    const unsigned char* row = src + ((size_t)y * w + x0) * 3;
    int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < n; ++t) {
        const int c = k[t];
        s0 += (int)row[3 * t + 0] * c;
        s1 += (int)row[3 * t + 1] * c;
        s2 += (int)row[3 * t + 2] * c;
    }
    unsigned char* o = tmp + (size_t)i * 3;
    o[0] = pil_clip8(s0); o[1] = pil_clip8(s1); o[2] = pil_clip8(s2);
}

// vertical pass + paste on the grey canvas + /255 + HWC -> CHW: tmp [th][tw][3] u8 -> dst [3][S][S] f32
__global__ __launch_bounds__(256) void letterbox_v_kernel(const unsigned char* __restrict__ tmp, int th, int tw,
                                                          float* __restrict__ dst, int S, int new_w, int new_h, int pad_left,
                                                          int pad_top, const int* __restrict__ yb, const int* __restrict__ yk,
                                                          int ksize, int vertical, int fill) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)S * S) return;
    const int cy = (int)(i / S), cx = (int)(i - (long long)cy * S);
    const int y = cy - pad_top, x = cx - pad_left;
    int v0 = fill, v1 = fill, v2 = fill;
    if ((unsigned)y < (unsigned)new_h && (unsigned)x < (unsigned)new_w) {
        if (vertical) {
            const int y0 = yb[2 * y], n = yb[2 * y + 1];
            const int* k = yk + (size_t)y * ksize;
            int s0 = 1 << (PIL_PRECISION_BITS - 1), s1 = s0, s2 = s0;
            for (int t = 0; t < n; ++t) {
                const unsigned char* p = tmp + ((size_t)(y0 + t) * tw + x) * 3;
                const int c = k[t];
                s0 += (int)p[0] * c; s1 += (int)p[1] * c; s2 += (int)p[2] * c;
            }
            v0 = pil_clip8(s0); v1 = pil_clip8(s1); v2 = pil_clip8(s2);
        } else {
            const unsigned char* p = tmp + ((size_t)y * tw + x) * 3;
            v0 = p[0]; v1 = p[1]; v2 = p[2];
        }
    }
    const size_t plane = (size_t)S * S;
    dst[i] = __fdiv_rn((float)v0, 255.0f);                  // torch: uint8 -> float32, / 255.0 (correctly rounded division)
    dst[plane + i] = __fdiv_rn((float)v1, 255.0f);
    dst[2 * plane + i] = __fdiv_rn((float)v2, 255.0f);
}

// nearest resize of the label map through host-built index tables + paste on a zero canvas + int64
__global__ __launch_bounds__(256) void letterbox_mask_kernel(const unsigned char* __restrict__ src, int w,
                                                             long long* __restrict__ dst, int S, int new_w, int new_h,
                                                             int pad_left, int pad_top, const int* __restrict__ xtab,
                                                             const int* __restrict__ ytab, int clip_max) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)S * S) return;
    const int cy = (int)(i / S), cx = (int)(i - (long long)cy * S);
    const int y = cy - pad_top, x = cx - pad_left;
    long long v = 0;
    if ((unsigned)y < (unsigned)new_h && (unsigned)x < (unsigned)new_w) {
        int m = src[(size_t)ytab[y] * w + xtab[x]];
        v = m > clip_max ? clip_max : m;                    // np.clip(mask, 0, num_classes - 1), :303
    }
    dst[i] = v;
}

extern "C" int ydl_letterbox_image(const void* src, int h, int w, void* tmp, float* dst, int S, int new_w, int new_h,
                                   int pad_left, int pad_top, const int* xbounds, const int* xcoef, int xksize,
                                   const int* ybounds, const int* ycoef, int yksize, int fill, void* stream) {
    YDL_CHECK(src && dst && h > 0 && w > 0 && S > 0, "bad image arguments");
    YDL_CHECK(new_w > 0 && new_h > 0 && pad_left >= 0 && pad_top >= 0 && pad_left + new_w <= S && pad_top + new_h <= S,
              "the resized image does not fit the canvas");
    const bool horiz = new_w != w, vert = new_h != h;
    YDL_CHECK(!horiz || (tmp && xbounds && xcoef && xksize > 0), "horizontal pass needs tmp and its coefficient tables");
    YDL_CHECK(!vert || (ybounds && ycoef && yksize > 0), "vertical pass needs its coefficient tables");
    hipStream_t st = (hipStream_t)stream;
    const unsigned char* mid = (const unsigned char*)src;
    if (horiz) {
        const long long n = (long long)h * new_w;
        letterbox_h_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>((const unsigned char*)src, h, w, (unsigned char*)tmp, new_w,
                                                                       xbounds, xcoef, xksize);
        mid = (const unsigned char*)tmp;
    }
    const long long n2 = (long long)S * S;
    letterbox_v_kernel<<<(unsigned)((n2 + 255) / 256), 256, 0, st>>>(mid, h, new_w, dst, S, new_w, new_h, pad_left, pad_top,
                                                                    ybounds, ycoef, yksize, vert ? 1 : 0, fill);
    YDL_LAUNCH_CHECK();
    return 0;
}

extern "C" int ydl_letterbox_mask(const void* src, int h, int w, int64_t* dst, int S, int new_w, int new_h, int pad_left,
                                  int pad_top, const int* xtab, const int* ytab, int clip_max, void* stream) {
    YDL_CHECK(src && dst && xtab && ytab && h > 0 && w > 0 && S > 0, "bad mask arguments");
    YDL_CHECK(new_w > 0 && new_h > 0 && pad_left >= 0 && pad_top >= 0 && pad_left + new_w <= S && pad_top + new_h <= S,
              "the resized mask does not fit the canvas");
    const long long n = (long long)S * S;
    letterbox_mask_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>((const unsigned char*)src, w, (long long*)dst, S,
                                                                                       new_w, new_h, pad_left, pad_top, xtab, ytab,
                                                                                       clip_max);
    YDL_LAUNCH_CHECK();
    return 0;
}
