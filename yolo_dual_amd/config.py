"""Process-wide settings of the HIP path."""
from __future__ import annotations

import os

_STATE = {"dtype": "bf16", "weight_epoch": 0, "lazy_upsample": True, "fuse_siblings": True, "overlap_wgrad": os.environ.get("YDL_OVERLAP_WGRAD", "1") != "0",
          "commute_concat": os.environ.get("YDL_COMMUTE_CONCAT", "1") != "0",
          "replicated_loss": os.environ.get("YDL_REPLICATED_LOSS", "1") != "0",
          "stem_s2d": os.environ.get("YDL_STEM_S2D", "1") != "0",
          # weight gradients: None = deterministic split-K (partial slabs + fixed-order sum) in f32 parity mode, f32 atomics in
          # bf16 throughput mode; True / False force one form for both
          "deterministic": {"1": True, "0": False}.get(os.environ.get("YDL_DETERMINISTIC", ""), None)}


def set_compute_dtype(name: str) -> None:
    """'bf16' (throughput mode: bf16 storage + bf16 MFMA, f32 accumulate/statistics) or 'f32' (parity mode: f32
    storage + exact f32 MFMA)."""
    if name not in ("bf16", "f32"):
        raise ValueError("compute dtype must be 'bf16' or 'f32'")
    _STATE["dtype"] = name


def compute_dtype() -> str:
    return _STATE["dtype"]


def lazy_upsample() -> bool:
    """nearest up-sampling by integer factors is kept lazy (point-wise consumers run at the low resolution)"""
    return _STATE["lazy_upsample"]


def set_lazy_upsample(on: bool) -> None:
    _STATE["lazy_upsample"] = bool(on)


def fuse_siblings() -> bool:
    """cv1/cv2 of a CSP block run as one convolution when their parameters are adjacent in the flat arenas"""
    return _STATE["fuse_siblings"]


def set_fuse_siblings(on: bool) -> None:
    _STATE["fuse_siblings"] = bool(on)


def overlap_wgrad() -> bool:
    """run weight-gradient kernels on a second stream, concurrently with the input-gradient chain"""
    return _STATE["overlap_wgrad"]


def wgrad_batch() -> int:
    """weight-gradient launches per fork of the side stream (tape._defer_wgrad)"""
    return _STATE.setdefault("wgrad_batch", max(int(os.environ.get("YDL_WGRAD_BATCH", "8")), 1))


def set_wgrad_batch(n: int) -> None:
    _STATE["wgrad_batch"] = max(int(n), 1)


def set_overlap_wgrad(on: bool) -> None:
    _STATE["overlap_wgrad"] = bool(on)


def commute_concat() -> bool:
    """conv1x1(cat(a, bilinear_up(b))) is evaluated as conv1x1_a(a) + bilinear_up(conv1x1_b(b)) (the wide low-resolution
    source is never up-sampled)"""
    return _STATE["commute_concat"]


def set_commute_concat(on: bool) -> None:
    _STATE["commute_concat"] = bool(on)


def resize_last() -> bool:
    """commuted conv over a virtual concat in replica-sums (throughput) mode: the plain sub-convolutions first, the resized source
    last through ydl_resize_acc_sums (read-modify-write + statistics in one streaming pass).  Off: the resized source initialises
    the output and the last sub-convolution accumulates with statistics (the only order of the partial-row / parity mode)"""
    # Default OFF — measured on MI355X, config 2, same-box A/B over 100 steps: 2362 / 2362 images/s with it against 2378 / 2372
    # without.  The streaming pass takes 70-85 us per call (not the ~45 us its bytes would take: ~130 VALU instructions per 16-byte
    # item — five unpacks, the bilinear blend, the statistics — make it instruction-bound), which is what the accumulate form of the
    # point-wise kernel costs over the plain one; the order buys nothing and stays as a tested alternative.
    return _STATE.setdefault("resize_last", os.environ.get("YDL_RESIZE_LAST", "0") != "0")


def set_resize_last(on: bool) -> None:
    _STATE["resize_last"] = bool(on)


def stem_s2d() -> bool:
    """a stem conv on the raw region input whose k and p are multiples of its stride runs as the equivalent stride-1 conv
    over the space-to-depth converted input"""
    return _STATE["stem_s2d"]


def set_stem_s2d(on: bool) -> None:
    _STATE["stem_s2d"] = bool(on)


def deterministic(dtype_name: str) -> bool:
    """weight gradients bitwise reproducible from run to run (ydl_conv_wgrad_det) — default: on in 'f32' parity mode"""
    d = _STATE["deterministic"]
    return (dtype_name == "f32") if d is None else bool(d)


def set_deterministic(on) -> None:
    """True / False, or None for the per-dtype default"""
    _STATE["deterministic"] = None if on is None else bool(on)


def bn_sums(dtype_name: str) -> bool:
    """BatchNorm statistics as atomically added replica sums (ydl_conv_fwd_sums / ydl_bn_act_fwd_sums / ydl_bn_act_bwd_sums: no
    finalize / merge launches).  Arrival-order sums: used where bitwise reproducibility is not asked for, i.e. together with the
    atomic weight gradients of throughput mode; YDL_BN_SUMS=0 / set_bn_sums(False) keeps the deterministic partial rows"""
    return _STATE.setdefault("bn_sums", os.environ.get("YDL_BN_SUMS", "1") != "0") and not deterministic(dtype_name)


def set_bn_sums(on: bool) -> None:
    _STATE["bn_sums"] = bool(on)


def bn_bwd_fuse() -> bool:
    """with replica sums: run the BatchNorm backward's reduce pass in the epilogue of the dgrad that writes the layer's output
    gradient last (ydl_conv_dgrad_bnred) instead of as its own launch.  OFF by default: measured on BASELINE config 2 the fused
    dgrads take 1.4 ms instead of 0.71 (the epilogue's SiLU' arithmetic lands on compute-bound kernels) for 0.27 ms less BatchNorm
    time, 2152 against 2245 images/s.  YDL_BN_FUSE=1 / set_bn_bwd_fuse(True) switches it on"""
    return _STATE.setdefault("bn_bwd_fuse", os.environ.get("YDL_BN_FUSE", "0") != "0")


def set_bn_bwd_fuse(on: bool) -> None:
    _STATE["bn_bwd_fuse"] = bool(on)


def sppf_argmax_hook():
    """test instrument (tests/test_gpu_model.py::test_bf16_tracks_f32): a callable that receives the three uint8 arg-max planes of an
    SPPF pool chain right after the forward launch and may copy other planes into them; None (default) = nothing is called"""
    return _STATE.get("sppf_argmax_hook")


def set_sppf_argmax_hook(fn) -> None:
    _STATE["sppf_argmax_hook"] = fn


def fuse_pw_backward() -> bool:
    """input + weight gradient of the HBM-bound 128 -> 128 1x1 layers in one pass over dy (ydl_conv_bwd_pw); throughput mode only"""
    return _STATE.get("fuse_pw_backward", os.environ.get("YDL_PWBW", "1") != "0")


def set_fuse_pw_backward(on: bool) -> None:
    _STATE["fuse_pw_backward"] = bool(on)


def set_dcnv3_border_rule(rule: str) -> None:
    """Which of the reference's two answers a DCNv3 sampling position EXACTLY at -1 gets (include/ydl.h: ydl_dcnv3_set_border_rule):
    "core" (default) = inside, as the pure-PyTorch core `dcnv3_core_pytorch` and the oracle have it (functions/dcnv3_func.py:148-189);
    "cuh" = outside, as the CUDA op tests it (dcnv3_im2col_cuda.cuh:262,334,428) — what a drop-in for the compiled extension selects."""
    from . import _lib as L
    if rule not in ("core", "cuh"):
        raise ValueError("rule must be 'core' or 'cuh'")
    L.lib().ydl_dcnv3_set_border_rule(1 if rule == "cuh" else 0)


def dcnv3_border_rule() -> str:
    from . import _lib as L
    return "cuh" if L.lib().ydl_dcnv3_get_border_rule() else "core"


def replicated_loss() -> bool:
    """SegmentationLoss evaluates a nearest-replicated prediction per stored pixel (ydl_seg_loss_rep_*)"""
    return _STATE["replicated_loss"]


def weight_epoch() -> int:
    return _STATE["weight_epoch"]


def bump_weight_epoch() -> None:
    """Called by optimizers that update parameter memory behind torch's version counters (the fused flat
    SGD+EMA kernel) so that Conv modules refresh their compute-layout weight copies."""
    _STATE["weight_epoch"] += 1


_GRAD_HOOKS = []


def add_grad_hook(fn) -> None:
    """fn(param) is called right after the kernels that write ``param.grad`` have been enqueued (data-parallel
    bucket triggering)."""
    _GRAD_HOOKS.append(fn)


def remove_grad_hook(fn) -> None:
    if fn in _GRAD_HOOKS:
        _GRAD_HOOKS.remove(fn)


def mark_touched(p) -> None:
    p._ydl_touched = True
    for fn in _GRAD_HOOKS:
        fn(p)
