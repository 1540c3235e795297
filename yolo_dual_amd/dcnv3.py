"""DCNv3 operator (grouped, mask-modulated deformable sampling, NHWC) on the HIP kernels.

Same call signature as the reference's ``DCNv3Function`` (models/ops_dcnv3/build/.../functions/dcnv3_func.py:19-61,
C++ side src/cuda/dcnv3_cuda.h:15-31): ``apply(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
dilation_h, dilation_w, group, group_channels, offset_scale, im2col_step)``.  ``im2col_step`` is accepted and ignored
(the HIP kernel has no batch chunking).  Gradients are accumulated in f32 like the reference (dcnv3_cuda.cu:126-133)."""
from __future__ import annotations

import ctypes

import torch

from . import _lib as L
from .tape import _p, _stream


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.YDL_F32
    if t.dtype == torch.bfloat16:
        return L.YDL_BF16
    raise TypeError("DCNv3 HIP kernels take float32 or bfloat16 tensors")


class DCNv3Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                group, group_channels, offset_scale, im2col_step=256):
        if input.device.type != "cuda":
            raise RuntimeError("DCNv3: Not implemented on the CPU (GPU only, no fallback)")
        input, offset, mask = input.contiguous(), offset.contiguous(), mask.contiguous()
        N, H, W, C = input.shape
        if C != group * group_channels:
            raise ValueError("input channels must equal group * group_channels")
        Ho = (H + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) // stride_h + 1
        Wo = (W + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) // stride_w + 1
        if tuple(offset.shape) != (N, Ho, Wo, group * kernel_h * kernel_w * 2):
            raise ValueError(f"offset must be (N, H_out, W_out, group*K*K*2) = {(N, Ho, Wo, group * kernel_h * kernel_w * 2)}")
        out = torch.empty((N, Ho, Wo, C), dtype=input.dtype, device=input.device)
        L.call("ydl_dcnv3_fwd", _dt(input), _p(input), _p(offset.to(input.dtype)), _p(mask.to(input.dtype)), _p(out),
               kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels,
               ctypes.c_float(offset_scale), N, H, W, Ho, Wo, _stream())
        ctx.save_for_backward(input, offset, mask)
        ctx.geom = (kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels,
                    float(offset_scale), N, H, W, Ho, Wo)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        input, offset, mask = ctx.saved_tensors
        (kh, kw, sh, sw, ph, pw, dh, dw, G, Gc, scale, N, H, W, Ho, Wo) = ctx.geom
        go = grad_output.contiguous().to(input.dtype)
        gin = torch.zeros(input.shape, dtype=torch.float32, device=input.device)
        goff = torch.empty(offset.shape, dtype=torch.float32, device=input.device)
        gmsk = torch.empty(mask.shape, dtype=torch.float32, device=input.device)
        L.call("ydl_dcnv3_bwd", _dt(input), _p(input), _p(offset.to(input.dtype)), _p(mask.to(input.dtype)), _p(go),
               _p(gin), _p(goff), _p(gmsk), kh, kw, sh, sw, ph, pw, dh, dw, G, Gc, ctypes.c_float(scale),
               N, H, W, Ho, Wo, _stream())
        return (gin.to(input.dtype), goff.to(offset.dtype), gmsk.to(mask.dtype)) + (None,) * 12


def dcnv3_core(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
               group, group_channels, offset_scale):
    """functional form with the argument list of ``dcnv3_core_pytorch`` (dcnv3_func.py:148-189)"""
    return DCNv3Function.apply(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h,
                               dilation_w, group, group_channels, offset_scale, 256)
