"""Spatial kernels driven through the C ABI (bilinear resize backward: the integer-factor form against the generic gather form)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("S,N,Hi,Wi,C,acc", [(4, 2, 10, 12, 24, 0), (2, 3, 7, 9, 16, 1), (4, 1, 40, 40, 128, 1), (2, 2, 1, 5, 8, 0)])
def test_integer_factor_bilinear_backward_equals_the_generic_gather(dtype, S, N, Hi, Wi, C, acc):
    """ydl_resize_bwd, bilinear / align_corners = False, output = S x input (the auto-align Concat of seg_diceloss_yolov5.py:484-507 in
    the live head: 40^2 -> 160^2): resize_bwd_int_kernel visits the 2S candidates per axis that can reference an input element, the
    generic kernel scans a conservative window — same index/weight arithmetic, same accumulation order: identical bits, incl. the
    clamped borders, one-row inputs and the accumulate form; and both equal torch's interpolate backward within rounding"""
    from yolo_dual_amd import _lib as L
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    dt = L.YDL_BF16 if dtype == "bf16" else L.YDL_F32
    gen = torch.Generator("cuda").manual_seed(S * 100 + C)
    Ho, Wo = S * Hi, S * Wi
    dy = torch.randn(N, Ho, Wo, C, device="cuda", generator=gen).to(tdt)
    dx0 = torch.randn(N, Hi, Wi, C, device="cuda", generator=gen).to(tdt)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    outs = []
    for fast in (1, 0):
        L.debug_set(10, fast)
        try:
            dx = dx0.clone()
            L.call("ydl_resize_bwd", dt, 1, P(dy), C, P(dx), C, acc, N, Hi, Wi, Ho, Wo, C, 0.0, 0.0, st)
            torch.cuda.synchronize()
            outs.append(dx)
        finally:
            L.debug_set(10, 1)
    assert torch.equal(outs[0], outs[1])
    x = torch.zeros(N, C, Hi, Wi, device="cuda", dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False)
    y.backward(dy.permute(0, 3, 1, 2).double())
    ref = x.grad.permute(0, 2, 3, 1) + (dx0.double() if acc else 0.0)
    err = float((outs[0].double() - ref).abs().max() / ref.abs().max())
    assert err < (1e-2 if dtype == "bf16" else 2e-6), err
