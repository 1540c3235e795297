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


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("mode,N,Hi,Wi,Ho,Wo,C", [(1, 2, 10, 12, 40, 48, 24), (1, 3, 7, 9, 14, 18, 16), (2, 1, 5, 6, 17, 11, 8), (0, 2, 4, 4, 8, 8, 40),
                                                  (1, 1, 40, 40, 160, 160, 128)])
def test_resize_accumulate_with_statistics(dtype, mode, N, Hi, Wi, Ho, Wo, C):
    """ydl_resize_acc_sums: y += resize(x) with the index arithmetic of ydl_resize_fwd (bit-equal to "resize into a buffer, then add in
    f32 and round once"), and the per-channel (sum, sum of squares) of the f32 results in the BatchNorm replica rows against a float64
    restatement (the last launch of the commuted Concat + 1x1 Conv, seg_diceloss_yolov5.py:484-507 + :388-409)"""
    from yolo_dual_amd import _lib as L
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    dt = L.YDL_BF16 if dtype == "bf16" else L.YDL_F32
    gen = torch.Generator("cuda").manual_seed(mode * 31 + C)
    x = torch.randn(N, Hi, Wi, C, device="cuda", generator=gen).to(tdt)
    y0 = torch.randn(N, Ho, Wo, C, device="cuda", generator=gen).to(tdt)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    Cp = (C + 7) // 8 * 8
    # reference: the resize in f32 through the existing kernel (f32 storage keeps the interpolated values unrounded)
    r32 = torch.zeros(N, Ho, Wo, C, device="cuda", dtype=torch.float32)
    L.call("ydl_resize_fwd", L.YDL_F32, mode, P(x.float().contiguous()), C, P(r32), C, N, Hi, Wi, Ho, Wo, C, 0.0, 0.0, st)
    want32 = y0.float() + r32
    y = y0.clone()
    sums = torch.zeros(L.BN_REPLICAS, 2, Cp, device="cuda")
    L.call("ydl_resize_acc_sums", dt, mode, P(x), C, P(y), C, N, Hi, Wi, Ho, Wo, C, 0.0, 0.0, P(sums), Cp, st)
    torch.cuda.synchronize()
    assert torch.equal(y, want32.to(tdt))
    got = sums.double().sum(0)[:, :C]
    ref = torch.stack([want32.double().sum((0, 1, 2)), (want32.double() ** 2).sum((0, 1, 2))])
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-5
    # without statistics: the same values, nothing else written
    y2 = y0.clone()
    L.call("ydl_resize_acc_sums", dt, mode, P(x), C, P(y2), C, N, Hi, Wi, Ho, Wo, C, 0.0, 0.0, None, 0, st)
    torch.cuda.synchronize()
    assert torch.equal(y2, y)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("mode,N,Hi,Wi,Ho,Wo,C", [(1, 2, 10, 12, 40, 48, 24), (2, 1, 5, 6, 17, 11, 8), (0, 2, 4, 4, 16, 16, 40), (1, 2, 40, 40, 20, 20, 64),
                                                  (1, 1, 3, 300, 7, 33, 8)])
def test_row_walking_resize_forward_equals_the_element_indexed_kernel(dtype, mode, N, Hi, Wi, Ho, Wo, C):
    """ydl_resize_fwd: the row-walking kernel (a CTA per output row, no 64-bit index decode per element) against the element-indexed
    kernel it replaces — same arithmetic per element, identical bits; nearest / bilinear / align_corners, up- and down-sampling"""
    from yolo_dual_amd import _lib as L
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    dt = L.YDL_BF16 if dtype == "bf16" else L.YDL_F32
    x = torch.randn(N, Hi, Wi, C, device="cuda", generator=torch.Generator("cuda").manual_seed(C + mode)).to(tdt)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    outs = []
    for rows in (1, 0):
        L.debug_set(11, rows)
        try:
            y = torch.full((N, Ho, Wo, C), 7.0, device="cuda", dtype=tdt)
            L.call("ydl_resize_fwd", dt, mode, P(x), C, P(y), C, N, Hi, Wi, Ho, Wo, C, 0.0, 0.0, st)
            torch.cuda.synchronize()
            outs.append(y)
        finally:
            L.debug_set(11, 1)
    assert torch.equal(outs[0], outs[1])


def test_given_scale_that_is_not_the_size_ratio_takes_the_generic_backward():
    """A caller-given scale (tape.resize passes the yaml Upsample's scale_factor; the C ABI takes scale_h / scale_w) need not equal
    Hi / Ho although the sizes look like an integer factor: scale_factor 2.05 on 12 columns gives floor(24.6) = 24 = 2 x 12, but output
    23 then maps to source 10.96 and references input 10 — outside the integer-factor kernel's fixed 2S window.  The dispatch must fall
    back to the generic gather; reference: torch's interpolate with the same scale_factor, float64."""
    from yolo_dual_amd import _lib as L
    sf = 2.05
    N, Hi, Wi, C = 2, 10, 12, 16
    Ho, Wo = int(Hi * sf), int(Wi * sf)
    assert (Ho, Wo) == (20, 24)
    gen = torch.Generator("cuda").manual_seed(11)
    dy = torch.randn(N, Ho, Wo, C, device="cuda", generator=gen)
    dx = torch.zeros(N, Hi, Wi, C, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    L.call("ydl_resize_bwd", L.YDL_F32, 1, P(dy), C, P(dx), C, 0, N, Hi, Wi, Ho, Wo, C, ctypes.c_float(1.0 / sf), ctypes.c_float(1.0 / sf), st)
    torch.cuda.synchronize()
    x = torch.zeros(N, C, Hi, Wi, device="cuda", dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.interpolate(x, scale_factor=sf, mode="bilinear", align_corners=False, recompute_scale_factor=False)
    assert tuple(y.shape[-2:]) == (Ho, Wo)
    y.backward(dy.permute(0, 3, 1, 2).double())
    ref = x.grad.permute(0, 2, 3, 1)
    err = float((dx.double() - ref).abs().max() / ref.abs().max())
    assert err < 5e-6, err
