"""The kernel instantiations that produce the benchmark number, pinned to the CPU oracle.

The golden/oracle block tests run at C <= 32 and H <= 36, where the launcher always picks the 64-pixel tiles.  Here ONE
``ydl.Conv`` (conv -> train-mode BN -> SiLU, seg_diceloss_yolov5.py:388-409) runs forward + backward at shapes that select
every instantiation on the benchmark's hot path — the 128-pixel implicit-GEMM tiles (3x3 and wide 1x1, 4 and 8 waves), the
weight-stationary point-wise streaming kernel for every K-row width, the strided one-launch dgrad, the 128-wide pipelined
weight-gradient kernel (M >= 200 000 pixels), deep split-K, and the two-level BN-statistics merge (> 1024 partial rows) —
and is compared with ``oracle.ref_cpu.conv_bn_act`` (pinned to the reference by tests/golden) on the same inputs.  Which
kernel ran is asserted through ``ydl_debug_last_kernel``.

Tolerances: f32 parity mode — output 1e-4, gradients 5e-4 (max-abs relative), as tests/test_gpu_blocks.py; bf16 throughput
mode — the oracle is run on the bf16-rounded input and weights, output 2e-2 and gradients 3e-2 in relative L2 (bf16 storage
of y / out / dy has 8 mantissa bits; a wrong tile index would give O(1))."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as R
from tests.util import l2_err, rel_err

pytestmark = pytest.mark.gpu


def _case(mode, N, c1, c2, k, s, H, W, seed=0, act=True, deterministic="keep"):
    """run Conv(c1, c2, k, s) fwd+bwd on the HIP path and on the CPU oracle; returns (got, ref, kernels)"""
    import yolo_dual_amd as ydl
    from yolo_dual_amd import _lib as L
    from yolo_dual_amd import config
    ydl.set_compute_dtype(mode)
    det_was = config._STATE["deterministic"]
    if deterministic != "keep":
        config.set_deterministic(deterministic)
    try:
        rs = np.random.RandomState(seed)
        fan_in = c1 * k * k
        w = torch.from_numpy((rs.standard_normal((c2, c1, k, k)) * np.sqrt(2.0 / fan_in)).astype(np.float32))
        gamma = torch.from_numpy(rs.uniform(0.5, 1.5, c2).astype(np.float32))
        beta = torch.from_numpy(rs.uniform(-0.3, 0.3, c2).astype(np.float32))
        x = torch.from_numpy((rs.standard_normal((N, c1, H, W)) + 0.25).astype(np.float32))
        if mode == "bf16":          # both sides see the same (bf16-representable) operands
            w = w.bfloat16().float()
            x = x.bfloat16().float()
        m = ydl.Conv(c1, c2, k, s, None, 1, act)
        with torch.no_grad():
            m.conv.weight.copy_(w)
            m.bn.weight.copy_(gamma)
            m.bn.bias.copy_(beta)
        m = m.cuda().train()
        xg = x.cuda().requires_grad_(True)
        out = m(xg)
        Ho, Wo = out.shape[2:]
        gup = torch.from_numpy(rs.standard_normal((N, c2, Ho, Wo)).astype(np.float32))
        (out * gup.cuda()).sum().backward()
        torch.cuda.synchronize()
        kern = {f: L.last_kernel(i) for i, f in enumerate(("fwd", "dgrad", "wgrad", "bn_finalize"))}
        got = dict(out=out.detach().cpu(), dx=xg.grad.detach().cpu(), dw=m.conv.weight.grad.detach().float().cpu(),
                   dgamma=m.bn.weight.grad.detach().cpu(), dbeta=m.bn.bias.grad.detach().cpu(),
                   rm=m.bn.running_mean.detach().cpu(), rv=m.bn.running_var.detach().cpu())
        # CPU oracle
        torch.set_num_threads(min(32, torch.get_num_threads() if torch.get_num_threads() > 1 else 16))
        sd = {"c.conv.weight": w.clone().requires_grad_(True), "c.bn.weight": gamma.clone().requires_grad_(True),
              "c.bn.bias": beta.clone().requires_grad_(True), "c.bn.running_mean": torch.zeros(c2),
              "c.bn.running_var": torch.ones(c2)}
        xr = x.clone().requires_grad_(True)
        o = R.conv_bn_act(sd, "c", xr, s=s, p=None, act="silu" if act else "none")
        (o * gup).sum().backward()
        ref = dict(out=o.detach(), dx=xr.grad, dw=sd["c.conv.weight"].grad, dgamma=sd["c.bn.weight"].grad,
                   dbeta=sd["c.bn.bias"].grad, rm=sd["c.bn.running_mean"], rv=sd["c.bn.running_var"])
        return got, ref, kern
    finally:
        config.set_deterministic(det_was)
        ydl.set_compute_dtype("bf16")


def _check(got, ref, mode, tag):
    if mode == "f32":
        assert rel_err(got["out"], ref["out"]) < 1e-4, (tag, "out", rel_err(got["out"], ref["out"]))
        for k in ("dx", "dw", "dgamma", "dbeta"):
            assert rel_err(got[k], ref[k]) < 5e-4, (tag, k, rel_err(got[k], ref[k]))
        for k in ("rm", "rv"):
            assert rel_err(got[k], ref[k]) < 1e-4, (tag, k, rel_err(got[k], ref[k]))
    else:
        assert l2_err(got["out"], ref["out"]) < 2e-2, (tag, "out", l2_err(got["out"], ref["out"]))
        for k in ("dx", "dw", "dgamma", "dbeta"):
            assert l2_err(got[k], ref[k]) < 3e-2, (tag, k, l2_err(got[k], ref[k]))
        for k in ("rm", "rv"):
            assert rel_err(got[k], ref[k]) < 1e-3, (tag, k, rel_err(got[k], ref[k]))   # statistics come from f32 accumulators


# (tag, mode, N, c1, c2, k, s, H, W, expected substrings of the fwd / dgrad / wgrad / bn_finalize kernel names)
TILED = [
    # 128x128 tile, 8 waves: the 3x3 layers of the 80x80 / 40x40 stages at bs 16 select it; same tile at bs 4 @160
    ("k3_128x128_f32", "f32", 4, 128, 128, 3, 1, 160, 160, ("igemm_kernel<f32,128,128,8", "igemm_kernel<f32,128,128,8", "wgrad_kernel<f32>", "")),
    ("k3_128x64_f32", "f32", 4, 64, 64, 3, 1, 160, 160, ("igemm_kernel<f32,128,64,4", "igemm_kernel<f32,128,64,4", "wgrad_kernel<f32>", "")),
    ("k3_512_40_f32", "f32", 16, 512, 512, 3, 1, 40, 40, ("igemm_kernel<f32,128,128,8", "igemm_kernel<f32,128,128,8", "wgrad_kernel<f32>", "")),
    # stride-2 down-sampler: forward gathers with in_mul 2, dgrad = 4 output-parity classes in one launch
    ("k3s2_128_256_f32", "f32", 4, 128, 256, 3, 2, 160, 160, ("igemm_kernel<f32,128,64,4", "igemm_kernel<f32,128,128,8", "wgrad_kernel<f32>", "")),
    # small-M deep layer: 64-pixel / 64-channel tiles
    ("k1_1024_20_f32", "f32", 16, 1024, 1024, 1, 1, 20, 20, ("igemm_kernel<f32,128,64,4", "igemm_kernel<f32,128,64,4", "wgrad_kernel<f32>", "")),
    # bf16: the throughput-mode instantiations of the same tiles (3x3 layers below 200 k pixels: the 128-wide weight-gradient kernel
    # sized to one wave of CTAs; 1x1 layers: the 64x64-tile kernel)
    # (>= 128 output channels and Cin % 64 == 0: the LDS-DMA ring kernel igemm2; otherwise igemm_kernel)
    # 3x3 / stride 1 with the image a multiple of 8 x 16: the patch-form kernel (activation patch loaded once per channel block)
    ("k3_128x128_bf16", "bf16", 4, 128, 128, 3, 1, 160, 160, ("igemm2h_kernel<128,128,2>", "igemm2h_kernel<128,128,2>", "wgrad3s_kernel<128>", "")),
    # ... the same layer on a 152 x 152 image (not a multiple of 16): the ring kernel
    ("k3_128x128_ring_bf16", "bf16", 4, 128, 128, 3, 1, 152, 152, ("igemm2l_kernel<256,128,8+4,3>", "igemm2l_kernel<256,128,8+4,3>", "wgrad3s_kernel<128>", "")),
    # ... at 3 x 128 x 128 (192 tiles of 128 x 128, 96 of 256 x 128): too few tiles for the one-CTA-per-CU form, the 128 x 128 / 64 x 128 ring
    ("k3_128x128_ring_small_bf16", "bf16", 3, 128, 128, 3, 1, 120, 136, ("igemm2l_kernel<256,128,8+4,3>", "igemm2l_kernel<256,128,8+4,3>", "", "")),
    # 64..127 stored output channels: the 128x64 ring tile (three CTAs per CU)
    ("k3_128x64_bf16", "bf16", 4, 128, 64, 3, 1, 160, 160, ("igemm2h_kernel<128,64,3>", "igemm2h_kernel<128,128,2>", "wgrad3s_kernel<64>", "")),
    ("k3_128x64_ring_bf16", "bf16", 4, 128, 64, 3, 1, 152, 152, ("igemm2_kernel<128,64,8,4,2>", "igemm2l_kernel<128,128,4+4,2>", "wgrad3s_kernel<64>", "")),
    # one channel block: the single-patch-buffer form (four CTAs per CU)
    ("k3_64x64_patch_bf16", "bf16", 8, 64, 64, 3, 1, 96, 160, ("igemm2h_kernel<128,64,2>", "igemm2h_kernel<128,64,2>", "wgrad3s_kernel<64>", "")),
    # 3x3 / s1 over ONE 64-channel block, >= 131 072 pixels, image width a multiple of 32: weights in registers (igemm2w_kernel, round 5)
    ("k3_64x64_wreg_bf16", "bf16", 8, 64, 64, 3, 1, 160, 160, ("igemm2w_kernel<64,nw4>", "igemm2w_kernel<64,nw4>", "wgrad3s_kernel<64>", "")),
    ("k3_64x128_wreg_bf16", "bf16", 6, 64, 128, 3, 1, 152, 160, ("igemm2w_kernel<128,nw4>", "igemm2h_kernel<128,64,3>", "wgrad3s_kernel<128>", "")),
    # k3 s2 p1 data gradient with the dy grid a multiple of 8 x 16: all four output-parity classes fused in one CTA (igemm2s_kernel)
    ("k3s2_64_128_bf16", "bf16", 4, 64, 128, 3, 2, 320, 320, ("igemm2l_kernel<128,128,4+4,2>", "igemm2s_kernel<128,64,2>", "wgrad3s_kernel<128>", "")),
    ("k3s2_128_256_fused_bf16", "bf16", 4, 128, 256, 3, 2, 160, 160, ("igemm2l_kernel<256,128,8+4,3>", "igemm2s_kernel<128,64,2>", "wgrad3s_kernel<128>", "")),
    # ... a dy grid of 88 x 88 (not a multiple of 16): the ring kernel, one launch over the four classes
    ("k3s2_64_128_ring_bf16", "bf16", 8, 64, 128, 3, 2, 176, 176, ("igemm2l_kernel<128,128,4+4,2>", "igemm2_kernel<128,64,8,4,2>", "", "")),
    # Cin not a multiple of 64: the register-staged kernel; its dgrad (96 output channels, K rows of 64) is ring-eligible
    ("k3_96_64_bf16", "bf16", 4, 96, 64, 3, 1, 160, 160, ("igemm_kernel<bf16,128,64,4", "igemm2h_kernel<128,64,2>", "wgrad3s_kernel<64>", "")),
    # (forward: 400 tiles of 256 x 128, 36 K-steps: the staggered one-CTA-per-CU form; the strided dgrad's parity classes stay on 128 x 128)
    ("k3s2_256_512_bf16", "bf16", 16, 256, 512, 3, 2, 80, 80, ("igemm2l_kernel<256,128,8+4,3>", "igemm2_kernel<128,128,8,4,2>", "wgrad3s_kernel<128>", "")),
    # small grids (< 256 tiles of 128x128): 64-pixel ring tiles; pixel-tile-fastest order for the 4.7 MB weight matrix
    ("k3_512_20_bf16", "bf16", 16, 512, 512, 3, 1, 20, 20, ("igemm2l_kernel<128,128,8+4,3>", "igemm2l_kernel<128,128,8+4,3>", "wgrad3s_kernel<128>", "")),
    ("k1_2048_1024_bf16", "bf16", 16, 2048, 1024, 1, 1, 20, 20, ("igemm2l_kernel<256,128,8+4,3>", "igemm2l_kernel<256,128,8+4,3>", "wgrad3s_kernel<128>", "")),
    # 12 -> 64 channels, 3x3 / s1 on a 16-channel-stride input: the thin-input kernel of the space-to-depth stem
    # (its weight gradient: the patch-form stemw_kernel from 65 536 pixels and image widths that are multiples of 64)
    ("stem_12_64_bf16", "bf16", 4, 12, 64, 3, 1, 320, 320, ("stem_kernel<bf16,16,64>", "", "stemw_kernel", "")),
    ("stem_12_64_small_bf16", "bf16", 2, 12, 64, 3, 1, 48, 64, ("stem_kernel<bf16,16,64>", "", "wgrad_kernel<bf16,tr>", "")),
    # ... 13 images of 50 x 128: two 64-pixel segments per row, a stage count (1300) that no CTA count divides
    ("stem_12_64_ragged_bf16", "bf16", 13, 12, 64, 3, 1, 50, 128, ("stem_kernel<bf16,16,64>", "", "stemw_kernel", "")),
    # ragged: 150 output channels (two channel tiles, the second one partial), odd image size, pixel tail
    ("k3_ragged_bf16", "bf16", 3, 64, 152, 3, 1, 75, 83, ("igemm2l_kernel<128,128,4+4,2>", "igemm_kernel<bf16,64,64,4", "wgrad3s_kernel<128>", "")),
]


@pytest.mark.parametrize("case", TILED, ids=[c[0] for c in TILED])
def test_tiled_igemm_hot_instantiations(case):
    tag, mode, N, c1, c2, k, s, H, W, exp = case
    got, ref, kern = _case(mode, N, c1, c2, k, s, H, W)
    for fam, e in zip(("fwd", "dgrad", "wgrad", "bn_finalize"), exp):
        assert kern[fam].startswith(e), (tag, fam, kern[fam], "expected", e)
    _check(got, ref, mode, tag)


# point-wise streaming kernel: K-row widths 128 / 256 / 512 bytes, 64 / 128 / 256 output channels, M >= 65 536 pixels
# (the last field of the recorded name is the store path: "ts" = LDS-transposed 16-byte stores, "direct" = register-layout stores)
PW = [
    ("pw_rb128_f32", "f32", 4, 32, 128, "pw_kernel<f32,128,8,4,direct>"),
    ("pw_rb256_f32", "f32", 4, 64, 128, "pw_kernel<f32,256,8,4,direct>"),
    ("pw_rb512_ct4_f32", "f32", 4, 128, 64, "pw_kernel<f32,512,4,4,direct>"),
    ("pw_rb512_nw8_f32", "f32", 4, 128, 128, "pw_kernel<f32,512,8,8,direct>"),
    ("pw_rb128_bf16", "bf16", 4, 64, 128, "pw_kernel<bf16,128,4,4,ts>"),
    ("pw_rb256_bf16", "bf16", 4, 128, 128, "pw_kernel<bf16,256,4,4,ts>"),
    ("pw_rb256_c256_bf16", "bf16", 4, 128, 256, "pw_kernel<bf16,256,4,4,ts>"),
    ("pw_rb128_c256_bf16", "bf16", 4, 64, 256, "pw_kernel<bf16,128,4,4,ts>"),
    ("pw_rb512_bf16", "bf16", 4, 256, 64, "pw_kernel<bf16,512,4,4,ts>"),
    # 256 -> 128: 512-byte rows on 64-channel waves
    ("pw_rb512_c128_bf16", "bf16", 4, 256, 128, "pw_kernel<bf16,512,4,4,ts>"),
    # 256 -> 256: the 8-wave instantiation (transposed stores since round 4: the reduction scratch aliases the store tiles, 128 KB of
    # weights + 32 KB of tiles = the CU's 160 KB; two pixel tiles in flight instead of three pay for the statistics' registers)
    ("pw_rb512_nw8_bf16", "bf16", 4, 256, 256, "pw_kernel<bf16,512,8,8,ts>"),
]


@pytest.mark.parametrize("case", PW, ids=[c[0] for c in PW])
def test_pointwise_streaming_kernel_against_oracle(case):
    tag, mode, N, c1, c2, expk = case
    got, ref, kern = _case(mode, N, c1, c2, 1, 1, 160, 160)
    assert kern["fwd"] == expk, (tag, kern)
    # the input gradient of a 1x1 conv is a 1x1 conv with Cin <-> Cout: also the streaming kernel when it qualifies
    _check(got, ref, mode, tag)


@pytest.mark.parametrize("c1,c2,k,expk", [(128, 128, 1, "pwbw_kernel<128,128>"), (128, 64, 3, "wgrad3s_kernel<64>"),
                                          (64, 128, 3, "wgrad3s_kernel<128>"), (128, 128, 1, "wgrad3s_kernel<128>")])
def test_wgrad2_pipelined_kernel_against_oracle(c1, c2, k, expk):
    """bf16, M = 8*160*160 = 204 800 >= 200 000 pixels: the 128-wide weight-gradient kernel (LDS-DMA feed, four loader waves: wgrad3s_kernel); the
    128 -> 128 1x1 layer as the model runs it — input and weight gradient from ONE pass over dy (pwbw_kernel, round 5) — and with
    that switched off (ydl_debug_set key 15): the separate weight-gradient launch"""
    from yolo_dual_amd import _lib as L
    two_launches = k == 1 and expk.startswith("wgrad3")
    if two_launches:
        L.debug_set(15, 0)
    try:
        got, ref, kern = _case("bf16", 8, c1, c2, k, 1, 160, 160)
    finally:
        L.debug_set(15, 1)
    assert kern["wgrad"] == expk, kern
    if expk.startswith("pwbw"):
        assert kern["dgrad"] == expk, kern
    _check(got, ref, "bf16", expk)


def test_bf16_dgrad_accumulate_variants_against_oracle():
    """the ACCUMULATING input-gradient launches of throughput mode (second writer of a shared gradient: read-modify-write
    epilogue of the ring kernel, 16-byte read-modify-write rows of the point-wise kernel) against the CPU oracle: an un-fused script C3 — cv1 and cv2
    read the same x, so cv1's dgrad (the later one in backward order) adds into what cv2's wrote; 128 channels at 4x160x160
    selects the streaming kernel for the 1x1 layers and the ring kernel for the 3x3"""
    import yolo_dual_amd as ydl
    from yolo_dual_amd import _lib as L
    from yolo_dual_amd import config
    from oracle.fill import fill_state_dict
    ydl.set_compute_dtype("bf16")
    config.set_fuse_siblings(False)
    try:
        m = ydl.C3(128, 128, 1)
        sd = m.state_dict()
        fill_state_dict(sd, 17, bn_stats=False)
        for k in sd:
            if sd[k].dim() == 4:
                sd[k] = sd[k].bfloat16().float()
        m.load_state_dict(sd)
        m = m.cuda().train()
        rs = np.random.RandomState(4)
        x = torch.from_numpy((rs.standard_normal((4, 128, 160, 160)) + 0.25).astype(np.float32)).bfloat16().float()
        gup = torch.from_numpy(rs.standard_normal((4, 128, 160, 160)).astype(np.float32))
        xg = x.cuda().requires_grad_(True)
        out = m(xg)
        (out * gup.cuda()).sum().backward()
        torch.cuda.synchronize()
        last_dgrad = L.last_kernel(1)
        # cv1's dgrad comes last and adds into what cv3's residual branch and cv2's dgrad wrote: the accumulating launch without
        # statistics goes through the LDS-transposed 16-byte read-modify-write rows
        assert last_dgrad.startswith("pw_kernel<bf16,") and last_dgrad.endswith(",ts>"), last_dgrad
        ps = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
        run = {k: v.clone() for k, v in sd.items()}
        run.update(ps)
        xr = x.clone().requires_grad_(True)
        torch.set_num_threads(min(32, torch.get_num_threads() if torch.get_num_threads() > 1 else 16))
    finally:
        config.set_fuse_siblings(True)
        ydl.set_compute_dtype("bf16")
    # the oracle keys blocks by prefix: wrap the block's state under "b."
    run = {"b." + k: v for k, v in run.items()}
    o = R.c3_script(run, "b", xr, 1, True)
    (o * gup).sum().backward()
    assert l2_err(out.detach().cpu(), o.detach()) < 3e-2, l2_err(out.detach().cpu(), o.detach())
    assert l2_err(xg.grad.detach().cpu(), xr.grad) < 5e-2, l2_err(xg.grad.detach().cpu(), xr.grad)
    named = dict(m.named_parameters())
    for k, p in ps.items():
        e = l2_err(named[k].grad.detach().float().cpu(), p.grad)
        assert e < 6e-2, (k, e)


@pytest.mark.parametrize("accumulate", [0, 1])
@pytest.mark.parametrize("cin,cout,N,Ho,Wo", [(64, 128, 2, 24, 32), (192, 64, 1, 16, 48), (72, 128, 2, 8, 16)])
def test_fused_stride2_dgrad_through_the_c_abi(cin, cout, N, Ho, Wo, accumulate):
    """ydl_conv_dgrad of a 3x3 / stride 2 / pad 1 convolution on the fused-parity kernel (igemm2s_kernel), overwrite and accumulate, against
    torch's conv2d input gradient in float64 on the same bf16 operands (seg_diceloss_yolov5.py:388-409 backward): image-border blocks,
    several channel blocks of dy, a partial last cin tile (192 = 3 x 64, 72 = 64 + 8) and the read-modify-write epilogue"""
    import ctypes
    from yolo_dual_amd import _lib as L
    rs = np.random.RandomState(cin + cout + accumulate)
    Hi, Wi = 2 * Ho, 2 * Wo
    dy = torch.from_numpy(rs.standard_normal((N, cout, Ho, Wo)).astype(np.float32)).bfloat16()
    w = torch.from_numpy((rs.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cout)).astype(np.float32)).bfloat16()
    dx0 = torch.from_numpy(rs.standard_normal((N, cin, Hi, Wi)).astype(np.float32)).bfloat16()
    ref = torch.nn.grad.conv2d_input((N, cin, Hi, Wi), w.double(), dy.double(), stride=2, padding=1)
    if accumulate:
        ref = ref + dx0.double()
    dev = torch.device("cuda")
    dy_g = dy.permute(0, 2, 3, 1).contiguous().to(dev)                       # NHWC
    wt_g = w.permute(1, 2, 3, 0).reshape(cin, 9, cout).contiguous().to(dev)   # [Cin][tap][Cout]
    dx_g = dx0.permute(0, 2, 3, 1).contiguous().to(dev)
    g = L.ConvGeom(N, Hi, Wi, cin, Ho, Wo, cout, 3, 2, 1, cin, cout)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    L.call("ydl_conv_dgrad", ctypes.byref(g), L.YDL_BF16, P(dy_g), P(wt_g), P(dx_g), accumulate, st)
    torch.cuda.synchronize()
    assert L.last_kernel(1) == ("igemm2s_kernel<128,64,2,acc>" if accumulate else "igemm2s_kernel<128,64,2>"), L.last_kernel(1)
    got = dx_g.float().permute(0, 3, 1, 2).cpu().double()
    # one bf16 rounding of an f32 accumulation: 2^-9 relative per element; judged in relative L2 and as a max error against the largest value
    assert l2_err(got, ref) < 4e-3, l2_err(got, ref)
    assert float((got - ref).abs().max()) < 1.6e-2 * float(ref.abs().max())


def test_bn_finalize_two_level_merge_against_oracle():
    """> 1024 per-block partial rows (147 456 pixels / 128): the level-1 pre-merge of ydl_bn_finalize"""
    got, ref, kern = _case("f32", 4, 16, 32, 3, 1, 192, 192)
    assert kern["bn_finalize"] == "bn_finalize<two-level>", kern
    _check(got, ref, "f32", "bn_two_level")


def test_parity_mode_weight_gradient_is_bitwise_reproducible():
    """YDL_F32 uses the deterministic split-K (partial slabs + fixed-order sum): two runs give identical bits, also with the
    weight-gradient kernels on the second stream (a cross-stream race would show up here as well)"""
    import yolo_dual_amd as ydl
    from yolo_dual_amd import config
    assert config.deterministic("f32") and not config.deterministic("bf16")
    res = []
    for _ in range(3):
        got, _ref, _k = _case("f32", 2, 64, 96, 3, 1, 96, 104, seed=3)
        res.append(got)
    for k in ("out", "dx", "dw", "dgamma", "dbeta"):
        assert torch.equal(res[0][k], res[1][k]) and torch.equal(res[0][k], res[2][k]), k
    # the atomic form of the same kernels agrees to rounding (same products, arrival-order sums)
    config.set_deterministic(False)
    try:
        got2, _ref, _k = _case("f32", 2, 64, 96, 3, 1, 96, 104, seed=3)
    finally:
        config.set_deterministic(None)
    assert rel_err(got2["dw"], res[0]["dw"]) < 1e-5
    # and the deterministic form of the bf16 kernels (incl. the 128-wide pipelined one) against their atomic form
    config.set_deterministic(True)
    try:
        a, _r, ka = _case("bf16", 8, 64, 64, 3, 1, 160, 160, seed=4)
        b, _r, _kb = _case("bf16", 8, 64, 64, 3, 1, 160, 160, seed=4)
    finally:
        config.set_deterministic(None)
    assert ka["wgrad"].startswith("wgrad3s_kernel")
    assert torch.equal(a["dw"], b["dw"])
    c, _r, _kc = _case("bf16", 8, 64, 64, 3, 1, 160, 160, seed=4)
    # (the throughput-mode forward of this layer is the weights-in-registers kernel, whose BatchNorm statistics are those of the STORED
    #  bf16 values — the tensor that is normalised — where the deterministic path sums the f32 accumulators: 2.6e-4 on dw, measured)
    assert rel_err(c["dw"], a["dw"]) < 6e-4


def test_launch_attributes_are_set_per_device():
    """kernel attributes (> 64 KB dynamic LDS) are keyed by the HIP device id: the first launch of an instantiation on a device
    registers it once; repeated launches on the same device do not"""
    import yolo_dual_amd as ydl
    from yolo_dual_amd import _lib as L
    ydl.set_compute_dtype("bf16")
    m = ydl.Conv(128, 128, 3, 1).cuda().train()
    x = torch.randn(2, 128, 96, 96, device="cuda")
    m(x)
    torch.cuda.synchronize()
    n1 = L.lib().ydl_debug_attr_sets()
    assert n1 >= 1
    m(x)
    torch.cuda.synchronize()
    assert L.lib().ydl_debug_attr_sets() == n1
    if torch.cuda.device_count() > 1:
        with torch.cuda.device(1):
            m1 = ydl.Conv(128, 128, 3, 1).to("cuda:1").train()
            out = m1(x.to("cuda:1"))
            torch.cuda.synchronize()
            assert torch.isfinite(out).all()
        assert L.lib().ydl_debug_attr_sets() > n1


# BatchNorm statistics as atomically added replica sums (throughput mode: ydl_conv_fwd_sums -> ydl_bn_act_fwd_sums,
# ydl_bn_act_bwd_sums; no finalize / merge launches).  Run in F32 with the non-deterministic switch so that the oracle's 1e-4 /
# 5e-4 bounds apply to the sums path itself (bf16 runs it by default: every bf16 case above goes through it).  One case per
# epilogue that adds into the sums: tiled kernel (3x3), point-wise streaming kernel, > 1024 blocks, Cout not a multiple of 8.
SUMS = [("sums_tiled_k3", 4, 64, 64, 3, 1, 96, 96), ("sums_pw", 4, 64, 128, 1, 1, 160, 160), ("sums_many_blocks", 4, 16, 32, 3, 1, 192, 192),
        ("sums_ragged_c", 3, 24, 20, 3, 2, 75, 83), ("sums_deep", 16, 256, 256, 1, 1, 20, 20)]


@pytest.mark.parametrize("case", SUMS, ids=[c[0] for c in SUMS])
def test_bn_replica_sums_against_oracle(case):
    tag, N, c1, c2, k, s_, H, W = case
    from yolo_dual_amd import config
    assert config.bn_sums("bf16") and not config.bn_sums("f32")
    got, ref, kern = _case("f32", N, c1, c2, k, s_, H, W, deterministic=False)
    assert config.bn_sums("bf16")
    _check(got, ref, "f32", tag)


def test_bn_replica_sums_in_blocks_against_the_deterministic_path():
    """split outputs (fused cv1|cv2 siblings: one sums buffer, two channel groups), residual joins and ReLU blocks: the replica-sums
    path (f32, non-deterministic switch) against the deterministic partial-row path of the same kernels, which the golden block
    tests pin to the reference"""
    import yolo_dual_amd as ydl
    from yolo_dual_amd import config
    from oracle.fill import fill_state_dict
    ydl.set_compute_dtype("f32")
    try:
        for make, shape in ((lambda: ydl.C3(64, 64, 1), (2, 64, 40, 44)), (lambda: ydl.C3(32, 48, 2, False), (2, 32, 33, 31)),
                            (lambda: ydl.C2f(64, 64, 1), (2, 64, 24, 24)), (lambda: ydl.BasicBlock(32, 32), (2, 32, 28, 28)),
                            (lambda: ydl.SPPF(64, 64), (2, 64, 20, 20))):
            res = []
            for det in (True, False):
                config.set_deterministic(det)
                m = make()
                sd = m.state_dict()
                fill_state_dict(sd, 11, bn_stats=True)
                m.load_state_dict(sd)
                m = m.cuda().train()
                opt = ydl.FlatSGDEMA(m, lr=0.01)           # adjacent arenas: the sibling pair runs fused
                x = torch.randn(*shape, device="cuda", generator=torch.Generator("cuda").manual_seed(2)).requires_grad_(True)
                opt.zero_grad()
                out = m(x)
                (out * torch.randn(out.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(3))).sum().backward()
                res.append([out.detach().cpu(), x.grad.cpu()] + [p.grad.detach().clone().cpu() for p in m.parameters()] +
                           [b.detach().clone().cpu() for b in m.buffers() if b.dtype.is_floating_point])
            for a, b in zip(*res):
                assert a.shape == b.shape and rel_err(b, a) < 2e-5, (shape, rel_err(b, a))
    finally:
        config.set_deterministic(None)
        ydl.set_compute_dtype("bf16")


@pytest.mark.parametrize("cin,cout,k,N,H,names", [
    (128, 128, 3, 4, 152, ("igemm2l_kernel<256,128,8+4,3>", "igemm2_kernel<256,128,8,4,3,stg>")),
    (256, 256, 3, 16, 40, ("igemm2l_kernel<256,128,8+4,3>", "igemm2_kernel<256,128,8,4,3,stg>")),
    (512, 512, 3, 16, 20, ("igemm2l_kernel<128,128,8+4,3>", "igemm2_kernel<64,128,4,2,3>")),
    (1024, 2048, 1, 16, 20, ("igemm2l_kernel<256,128,8+4,3>", "igemm2_kernel<256,128,8,4,3,stg>")),
    (136, 192, 3, 2, 83, ("igemm2l_kernel<128,128,8+4,3>", "igemm2_kernel<64,128,4,2,3>")),
    (512, 512, 1, 16, 40, ("igemm2l_kernel<128,128,4+4,2>", "igemm2_kernel<128,128,8,4,2>")),
    (136, 192, 3, 3, 83, ("igemm2l_kernel<128,128,4+4,2>", "igemm2_kernel<128,128,8,4,2>"))])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_loader_wave_ring_kernel_equals_the_plain_ring_kernel(cin, cout, k, N, H, names, accumulate):
    """igemm2l_kernel (eight waves only multiply, four only issue the LDS-DMA ring: ydl_debug_set key 19) against igemm2_kernel (every
    wave does both) through ydl_conv_dgrad: the same K-step order into the same accumulators, so the two are equal bit for bit — 3x3
    tiles with padding taps and pixel tails, a deep 1x1, the small-grid 128 x 128 form, the two-CTA four + four wave form, ragged maps with
    partial channel tiles,
    overwrite and gradient fan-in — and both against float64 on the same bf16 operands (seg_diceloss_yolov5.py:388-409 backward)"""
    import ctypes
    from yolo_dual_amd import _lib as L
    rs = np.random.RandomState(cin + cout + k)
    p_ = k // 2
    dy = torch.from_numpy(rs.standard_normal((N, cout, H, H)).astype(np.float32)).bfloat16()
    w = torch.from_numpy((rs.standard_normal((cout, cin, k, k)) / np.sqrt(k * k * cout)).astype(np.float32)).bfloat16()
    dx0 = torch.from_numpy(rs.standard_normal((N, cin, H, H)).astype(np.float32)).bfloat16()
    ref = torch.nn.grad.conv2d_input((N, cin, H, H), w.float(), dy.float(), stride=1, padding=p_).double() + (dx0.double() if accumulate else 0.0)
    dev = torch.device("cuda")
    dy_g = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    wt_g = w.permute(1, 2, 3, 0).reshape(cin, k * k, cout).contiguous().to(dev)       # [Cin][tap][Cout]
    g = L.ConvGeom(N, H, H, cin, H, H, cout, k, 1, p_, cin, cout)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    res, kerns = [], []
    L.debug_set(8, 0)              # (the 3x3 shapes on the ring family, not on the patch-form kernels)
    try:
        for on in (1, 0):
            L.debug_set(19, on)
            dx_g = dx0.permute(0, 2, 3, 1).contiguous().to(dev)
            L.call("ydl_conv_dgrad", ctypes.byref(g), L.YDL_BF16, P(dy_g), P(wt_g), P(dx_g), accumulate, st)
            torch.cuda.synchronize()
            kerns.append(L.last_kernel(1))
            res.append(dx_g)
    finally:
        L.debug_set(19, -1)
        L.debug_set(8, 1)
    if names is not None:
        assert tuple(kerns) == names, kerns
    assert torch.equal(res[0], res[1]), rel_err(res[0].float(), res[1].float())
    got = res[0].float().permute(0, 3, 1, 2).cpu().double()
    assert l2_err(got, ref) < 4e-3, l2_err(got, ref)


@pytest.mark.parametrize("shape", [(8, 128, 128, 1, 1, 160), (8, 128, 64, 3, 1, 160), (4, 64, 128, 3, 2, 320), (16, 256, 512, 3, 2, 80),
                                   (3, 64, 152, 3, 1, 83)])
def test_wgrad_lds_dma_feed_equals_register_staged_kernel(shape):
    """three feeds of the same tile: wgrad3s_kernel (four loader waves DMA the operands straight into the swizzled LDS image, the
    other four only multiply — the default), wgrad3_kernel (every wave loads and multiplies; one, two loader waves as well) and
    wgrad2_kernel (global -> registers -> ds_write): same tiles, same stage order, same MFMA chains — in the deterministic slab form all
    are equal bit for bit, on 256-byte and 128-byte dY rows, stride 2, pixel tails and a partial channel tile"""
    from yolo_dual_amd import _lib as L
    N, c1, c2, k, s_, H = shape
    res = []
    try:
        for dma, loaders in ((1, 4), (1, 0), (1, 2), (1, 1), (0, 0)):
            L.debug_set(4, dma)
            L.debug_set(18, loaders)
            got, _ref, kern = _case("bf16", N, c1, c2, k, s_, H, H, seed=5, deterministic=True)
            assert kern["wgrad"].startswith(("wgrad3s_kernel" if loaders else "wgrad3_kernel") if dma else "wgrad2_kernel"), kern
            res.append(got["dw"])
    finally:
        L.debug_set(4, 1)
        L.debug_set(18, -1)
    for r in res[1:]:
        assert torch.equal(res[0], r), rel_err(res[0], r)


@pytest.mark.parametrize("dtype,N,C,H,W,k", [("bf16", 3, 72, 20, 20, 5), ("f32", 2, 20, 13, 17, 5), ("bf16", 2, 16, 9, 31, 3),
                                             ("bf16", 16, 512, 20, 20, 5)])
def test_sppf_pool_chain_in_one_launch_equals_three_maxpools(dtype, N, C, H, W, k):
    """ydl_sppf_pool_fwd / _bwd (SPPF's three chained 5x5 pools with the plane held in LDS, seg_diceloss_yolov5.py:468-481) against
    three ydl_maxpool_fwd / ydl_maxpool_bwd launches on concat-slice layouts: outputs, arg-max planes and the gradient that
    reaches x are equal bit for bit — bf16 activations are full of exact ties, so the first-maximum rule is exercised"""
    import ctypes
    from yolo_dual_amd import _lib as L
    dt = L.YDL_BF16 if dtype == "bf16" else L.YDL_F32
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    V = 8 if dtype == "bf16" else 4
    Cp = (C + V - 1) // V * V
    ld = 4 * Cp
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    assert L.lib().ydl_sppf_pool_supported(dt, H, W, C, k) == 1
    gen = torch.Generator("cuda").manual_seed(7)
    res = []
    for fused in (True, False):
        cat = torch.zeros(N, H, W, ld, device="cuda", dtype=tdt)
        cat[..., :C] = (torch.randn(N, H, W, C, device="cuda", generator=torch.Generator("cuda").manual_seed(7)) * 2).round().to(tdt) / 2
        sl = [cat[..., i * Cp:] for i in range(4)]
        idx = [torch.zeros(N * H * W * Cp, dtype=torch.uint8, device="cuda") for _ in range(3)]
        dcat = torch.randn(N, H, W, ld, device="cuda", generator=torch.Generator("cuda").manual_seed(9)).to(tdt)
        dsl = [dcat[..., i * Cp:] for i in range(4)]
        if fused:
            L.call("ydl_sppf_pool_fwd", dt, P(sl[0]), ld, P(sl[1]), P(sl[2]), P(sl[3]), ld, P(idx[0]), P(idx[1]), P(idx[2]), N, H, W, C, k, st)
            L.call("ydl_sppf_pool_bwd", dt, P(dsl[1]), P(dsl[2]), P(dsl[3]), ld, P(idx[0]), P(idx[1]), P(idx[2]), P(dsl[0]), ld, 1,
                   N, H, W, C, k, st)
        else:
            for i in range(3):
                L.call("ydl_maxpool_fwd", dt, P(sl[i]), ld, P(sl[i + 1]), ld, P(idx[i]), N, H, W, H, W, C, k, 1, k // 2, st)
            for i in (2, 1, 0):
                L.call("ydl_maxpool_bwd", dt, P(dsl[i + 1]), ld, P(idx[i]), P(dsl[i]), ld, 1, N, H, W, H, W, C, k, 1, k // 2, st)
        torch.cuda.synchronize()
        res.append((cat.float().cpu(), [i.cpu() for i in idx], dcat[..., :Cp].float().cpu()))
    assert torch.equal(res[0][0], res[1][0])
    assert float(res[0][0][..., Cp:2 * Cp].abs().max()) > 0          # the pools did write
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    assert torch.equal(res[0][2], res[1][2])


BNRED = [("ring9_two_segments", 4, 128, 128, 3, 1, 80, 80, [(0, 64, 1), (64, 128, 0)], 0),
         ("ring13_one_segment_accumulate", 8, 64, 128, 3, 1, 160, 160, [(0, 64, 1)], 1),
         ("ring7_partial_cover", 16, 128, 512, 1, 1, 48, 48, [(64, 128, 1)], 0),
         ("ring13_stride2_classes", 4, 64, 128, 3, 2, 160, 160, [(0, 64, 1)], 0),
         ("ring7_persistent", 16, 128, 64, 3, 1, 160, 160, [(0, 128, 1)], 0)]


@pytest.mark.parametrize("case", BNRED, ids=[c[0] for c in BNRED])
def test_dgrad_with_fused_bn_backward_reduce(case):
    """ydl_conv_dgrad_bnred: the input gradient equals ydl_conv_dgrad's bit for bit, and the per-channel (sum dz, sum dz*xhat) the
    epilogue adds into the replica rows equal the BatchNorm-backward reduce of that stored gradient (float64 restatement of
    bn_bwd_reduce: dz = dout * silu'(y*scale + shift), xhat = (y - mean) * invstd) — one and two producer segments, accumulate,
    a segment covering half of the channels, the stride-2 parity classes, the persistent ring kernel"""
    import ctypes
    from yolo_dual_amd import _lib as L
    tag, N, Cin, Cout, k, s_, Hi, Wi, segs, acc = case
    p = k // 2
    Ho, Wo = (Hi + 2 * p - k) // s_ + 1, (Wi + 2 * p - k) // s_ + 1
    dev = "cuda"
    gen = torch.Generator(dev).manual_seed(11)
    rnd = lambda *sh: torch.randn(*sh, device=dev, generator=gen)
    dy = rnd(N, Ho, Wo, Cout).bfloat16()
    wt = (rnd(Cin, k * k, Cout) * 0.05).bfloat16()
    g = L.ConvGeom(N, Hi, Wi, Cin, Ho, Wo, Cout, k, s_, p, Cin, Cout, 0)
    gp = ctypes.byref(g)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    assert L.lib().ydl_conv_dgrad_bnred_supported(gp, L.YDL_BF16) == 1, tag
    base = (rnd(N, Hi, Wi, Cin) * 0.5).bfloat16() if acc else torch.zeros(N, Hi, Wi, Cin, device=dev, dtype=torch.bfloat16)
    dx_ref = base.clone()
    L.debug_set(8, 0)          # the reference launch on the ring kernel too (the patch-form kernels have no fused epilogue)
    L.debug_set(9, 0)
    L.debug_set(17, 0)         # (nor has the weights-in-registers kernel, nor the loader-wave ring)
    L.debug_set(19, 0)
    try:
        L.call("ydl_conv_dgrad", gp, L.YDL_BF16, P(dy), P(wt), P(dx_ref), acc, st)
        plain = L.last_kernel(1)
    finally:
        L.debug_set(8, 1)
        L.debug_set(9, 1)
        L.debug_set(17, 1)
        L.debug_set(19, -1)
    red = L.BnRed()
    red.nseg = len(segs)
    keep = []
    for i, (c0, c1, act) in enumerate(segs):
        cw = c1 - c0
        y = rnd(N, Hi, Wi, cw).bfloat16()
        sc, sf = rnd(cw) * 0.5 + 1.0, rnd(cw) * 0.3
        mu, inv = rnd(cw) * 0.2, rnd(cw).abs() + 0.5
        sums = torch.zeros(L.BN_REPLICAS * 2 * cw, device=dev)
        red.c0[i], red.c1[i], red.ldy[i], red.cp[i], red.act[i] = c0, c1, cw, cw, act
        red.y[i], red.scale[i], red.shift[i] = y.data_ptr(), sc.data_ptr(), sf.data_ptr()
        red.mean[i], red.invstd[i], red.sums[i] = mu.data_ptr(), inv.data_ptr(), sums.data_ptr()
        keep.append((y, sc, sf, mu, inv, sums))
    dx = base.clone()
    L.call("ydl_conv_dgrad_bnred", gp, L.YDL_BF16, P(dy), P(wt), P(dx), acc, ctypes.byref(red), st)
    fusedk = L.last_kernel(1)
    torch.cuda.synchronize()
    assert "bnred" in fusedk and fusedk.replace(",bnred", "") == plain.replace(":persistent", ""), (plain, fusedk)
    if tag == "ring7_persistent":
        assert plain.endswith(":persistent"), plain        # (the fused form runs one tile per CTA: register budget)
    assert torch.equal(dx, dx_ref)
    for (c0, c1, act), (y, sc, sf, mu, inv, sums) in zip(segs, keep):
        cw = c1 - c0
        d = dx_ref[..., c0:c1].double().reshape(-1, cw)
        yy = y.double().reshape(-1, cw)
        z = yy * sc.double() + sf.double()
        sg = torch.sigmoid(z)
        dz = d * (sg * (1 + z * (1 - sg))) if act else d
        xh = (yy - mu.double()) * inv.double()
        ref = torch.stack([dz.sum(0), (dz * xh).sum(0)])
        got = sums.view(L.BN_REPLICAS, 2, cw).double().sum(0)
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < 2e-4, (tag, c0, err)


@pytest.mark.parametrize("acc_path", [1, 2])          # ydl_debug_set key 14: 1 = transposed stores + row-layout statistics, 2 = direct stores
@pytest.mark.parametrize("cin,cout,N,H,W,expect", [(128, 128, 4, 128, 128, "pw_kernel<bf16,256,4,4,"), (256, 256, 4, 128, 128, "pw_kernel<bf16,512,8,8,"),
                                                    (64, 128, 5, 120, 112, "pw_kernel<bf16,128,4,4,")])
def test_accumulating_pointwise_forward_with_statistics_through_the_c_abi(cin, cout, N, H, W, expect, acc_path):
    """ydl_conv_fwd_sums(accumulate = 1) on the point-wise streaming kernel: y += W x with the BatchNorm (sum, sum of squares) replica rows of
    the FINAL y — the second launch of the commuted Concat + 1x1 Conv of the live head (seg_diceloss_yolov5.py:484-507 + :388-409), the
    instantiations bench.py runs (`pw_kernel<bf16,256,4,4,*,acc,stats>`, `<bf16,512,8,8,*,acc,stats>`).  Reference: float64 on the same
    bf16 operands.  Both store paths: the per-wave transposed one (round 5: statistics taken in the row layout; its own contribution is
    rounded to bf16 before the add — one rounding more) and the direct read-modify-write one."""
    import ctypes
    from yolo_dual_amd import _lib as L
    rs = np.random.RandomState(cin + cout + acc_path)
    x = torch.from_numpy((rs.standard_normal((N, H, W, cin)) + 0.25).astype(np.float32)).bfloat16()
    w = torch.from_numpy((rs.standard_normal((cout, cin)) / np.sqrt(cin)).astype(np.float32)).bfloat16()
    y0 = torch.from_numpy(rs.standard_normal((N, H, W, cout)).astype(np.float32)).bfloat16()
    ref = y0.double() + x.double() @ w.double().t()
    dev = torch.device("cuda")
    xg, wg, yg = x.to(dev), w.view(cout, 1, cin).contiguous().to(dev), y0.clone().to(dev)
    sums = torch.zeros(8, 2, cout, device=dev)                       # YDL_BN_REPLICAS rows of (sum, sum of squares)
    g = L.ConvGeom(N, H, W, cin, H, W, cout, 1, 1, 0, cin, cout)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    L.debug_set(14, acc_path)
    try:
        L.call("ydl_conv_fwd_sums", ctypes.byref(g), L.YDL_BF16, P(xg), P(wg), P(yg), P(sums), 1, st)
        torch.cuda.synchronize()
        name = L.last_kernel(0)
    finally:
        L.debug_set(14, -1)
    assert name.startswith(expect) and name.endswith(",ts>" if acc_path == 1 else ",direct>"), name
    got = yg.cpu().double()
    assert l2_err(got, ref) < 6e-3, l2_err(got, ref)
    assert float((got - ref).abs().max()) < 2.5e-2 * float(ref.abs().max())
    tot = sums.double().sum(0).cpu()
    r1, r2 = ref.reshape(-1, cout).sum(0), (ref * ref).reshape(-1, cout).sum(0)
    # the sums are of the f32 values before the final rounding (transposed path: own contribution already bf16-rounded)
    # (a sum of n values carrying independent 2^-9 roundings is off by about 2^-9 x rms x sqrt(n) = 2^-9 x sqrt(sum of squares))
    assert float(((tot[0] - r1).abs() / r2.sqrt()).max()) < 5e-3, float(((tot[0] - r1).abs() / r2.sqrt()).max())
    assert float(((tot[1] - r2) / r2).abs().max()) < 2e-3, float(((tot[1] - r2) / r2).abs().max())


@pytest.mark.parametrize("cin,cout,N,H,W,expect", [(128, 128, 4, 96, 96, "igemm2h_kernel<128,128,2>"), (64, 64, 3, 24, 32, "igemm2h_kernel<128,64,2>"),
                                                    (64, 128, 2, 40, 32, "igemm2h_kernel<128,64,3>")])
def test_accumulating_patch_form_dgrad_through_the_c_abi(cin, cout, N, H, W, expect):
    """ydl_conv_dgrad(accumulate = 1) of a 3x3 / stride 1 / pad 1 convolution on the patch-form kernels (the read-modify-write pre-pass of
    igemm2_epilogue behind igemm2hs_kernel — two patch buffers, the one-channel-block form — and igemm2h_kernel<64, 3 stages>), which
    the whole-model tests reach in f32 only: float64 reference on the same bf16 operands (seg_diceloss_yolov5.py:388-409 backward)."""
    import ctypes
    from yolo_dual_amd import _lib as L
    rs = np.random.RandomState(cin * 3 + cout)
    dy = torch.from_numpy(rs.standard_normal((N, cout, H, W)).astype(np.float32)).bfloat16()
    w = torch.from_numpy((rs.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cout)).astype(np.float32)).bfloat16()
    dx0 = torch.from_numpy(rs.standard_normal((N, cin, H, W)).astype(np.float32)).bfloat16()
    ref = torch.nn.grad.conv2d_input((N, cin, H, W), w.double(), dy.double(), stride=1, padding=1) + dx0.double()
    dev = torch.device("cuda")
    dy_g = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    wt_g = w.permute(1, 2, 3, 0).reshape(cin, 9, cout).contiguous().to(dev)       # [Cin][tap][Cout]
    dx_g = dx0.permute(0, 2, 3, 1).contiguous().to(dev)
    g = L.ConvGeom(N, H, W, cin, H, W, cout, 3, 1, 1, cin, cout)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    L.call("ydl_conv_dgrad", ctypes.byref(g), L.YDL_BF16, P(dy_g), P(wt_g), P(dx_g), 1, st)
    torch.cuda.synchronize()
    assert L.last_kernel(1) == expect, L.last_kernel(1)
    got = dx_g.float().permute(0, 3, 1, 2).cpu().double()
    assert l2_err(got, ref) < 4e-3, l2_err(got, ref)
    assert float((got - ref).abs().max()) < 1.6e-2 * float(ref.abs().max())


@pytest.mark.parametrize("accumulate", [0, 1])
@pytest.mark.parametrize("N,H,W,ldx,ldy,lddx,ldw", [(2, 256, 256, 128, 128, 128, 0), (3, 211, 209, 192, 128, 256, 640), (1, 363, 365, 128, 136, 128, 0)])
def test_one_pass_pointwise_backward_through_the_c_abi(N, H, W, ldx, ldy, lddx, ldw, accumulate):
    """ydl_conv_bwd_pw: input gradient and weight gradient of a 128 -> 128 1x1 convolution in one pass over dy (pwbw_kernel; the five
    HBM-bound 160^2 layers of config 2) against float64 on the same bf16 operands (seg_diceloss_yolov5.py:388-409 backward): pixel counts
    that no stage / CTA count divides, channel slices of wider buffers on every operand (x, dy, dx) and a wider weight-gradient row
    (the column block of a commuted Concat + Conv), overwrite and gradient fan-in; dW receives atomic adds on top of its contents."""
    import ctypes
    from yolo_dual_amd import _lib as L
    rs = np.random.RandomState(N * 7 + H + accumulate)
    M = N * H * W
    C = 128
    x = torch.from_numpy(rs.standard_normal((M, C)).astype(np.float32)).bfloat16()
    dy = torch.from_numpy(rs.standard_normal((M, C)).astype(np.float32)).bfloat16()
    w = torch.from_numpy((rs.standard_normal((C, C)) / np.sqrt(C)).astype(np.float32)).bfloat16()          # [co][ci]
    dx0 = torch.from_numpy(rs.standard_normal((M, C)).astype(np.float32)).bfloat16()
    dw0 = torch.from_numpy(rs.standard_normal((C, C)).astype(np.float32))
    ref_dx = dy.double() @ w.double() + (dx0.double() if accumulate else 0.0)
    ref_dw = dw0.double() + dy.double().t() @ x.double()
    dev = torch.device("cuda")
    canary = 7.0
    xg = torch.full((M, ldx), canary, dtype=torch.bfloat16, device=dev); xg[:, :C] = x.to(dev)
    dyg = torch.full((M, ldy), canary, dtype=torch.bfloat16, device=dev); dyg[:, :C] = dy.to(dev)
    dxg = torch.full((M, lddx), canary, dtype=torch.bfloat16, device=dev); dxg[:, :C] = dx0.to(dev)
    ldw_e = ldw or C
    dwg = torch.full((C, ldw_e), canary, dtype=torch.float32, device=dev); dwg[:, :C] = dw0.to(dev)
    wt = w.t().contiguous().to(dev)                                   # [ci][co]
    g = L.ConvGeom(N, H, W, C, H, W, C, 1, 1, 0, ldx, ldy, ldw)
    assert L.lib().ydl_conv_bwd_pw_supported(ctypes.byref(g), L.YDL_BF16) == 1
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    L.call("ydl_conv_bwd_pw", ctypes.byref(g), L.YDL_BF16, P(xg), P(dyg), P(wt), P(dxg), lddx, accumulate, P(dwg), st)
    torch.cuda.synchronize()
    assert L.last_kernel(1) == ("pwbw_kernel<128,128,acc>" if accumulate else "pwbw_kernel<128,128>") and L.last_kernel(2) == "pwbw_kernel<128,128>"
    got_dx = dxg[:, :C].float().cpu().double()
    assert l2_err(got_dx, ref_dx) < 4e-3, l2_err(got_dx, ref_dx)
    assert float((got_dx - ref_dx).abs().max()) < 1.6e-2 * float(ref_dx.abs().max())
    got_dw = dwg[:, :C].cpu().double()
    assert l2_err(got_dw, ref_dw) < 1e-5, l2_err(got_dw, ref_dw)          # f32 accumulation of exact bf16 products
    # nothing outside the logical channels was written
    for buf, ld in ((dxg, lddx), (dwg, ldw_e)):
        if ld > C:
            assert bool((buf[:, C:].float() == canary).all())
    # smaller maps keep the two-launch form
    g2 = L.ConvGeom(1, 64, 64, C, 64, 64, C, 1, 1, 0, C, C, 0)
    assert L.lib().ydl_conv_bwd_pw_supported(ctypes.byref(g2), L.YDL_BF16) == 0
