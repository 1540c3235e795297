"""ctypes binding of libydl_hip.so (include/ydl.h).  The product path has NO fallback: if the library is missing
or a call fails, we raise."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YDL_LIB", os.path.join(_HERE, "lib", "libydl_hip.so"))   # YDL_LIB: dev override for A/B builds

YDL_F32, YDL_BF16, YDL_F16 = 0, 1, 2
BN_REPLICAS = 8             # YDL_BN_REPLICAS of ydl.h
ACT_NONE, ACT_SILU, ACT_RELU = 0, 1, 2
RES_NONE, RES_AFTER_ACT, RES_BEFORE_ACT = 0, 1, 2
RES_GRAD_ACCUMULATE = 16
LOSS_DICE, LOSS_JACCARD = 0, 1
RESIZE_NEAREST, RESIZE_BILINEAR, RESIZE_BILINEAR_AC = 0, 1, 2


class ConvGeom(C.Structure):
    _fields_ = [("N", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int), ("Cin", C.c_int),
                ("Ho", C.c_int), ("Wo", C.c_int), ("Cout", C.c_int),
                ("k", C.c_int), ("s", C.c_int), ("p", C.c_int),
                ("ldx", C.c_int), ("ldy", C.c_int), ("ldw", C.c_int)]


class BnRed(C.Structure):
    """ydl_bnred of ydl.h: up to two channel segments of a gradient whose BatchNorm-backward reduce pass runs in the dgrad epilogue"""
    _fields_ = [("nseg", C.c_int), ("c0", C.c_int * 2), ("c1", C.c_int * 2), ("ldy", C.c_int * 2), ("cp", C.c_int * 2), ("act", C.c_int * 2),
                ("y", C.c_void_p * 2), ("scale", C.c_void_p * 2), ("shift", C.c_void_p * 2), ("mean", C.c_void_p * 2),
                ("invstd", C.c_void_p * 2), ("sums", C.c_void_p * 2)]


_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
_G = C.POINTER(ConvGeom)
_R = C.POINTER(BnRed)

# name -> (restype, argtypes); mirrors include/ydl.h one to one
SIGNATURES = {
    "ydl_last_error": (C.c_char_p, []),
    "ydl_version": (_i, []),
    "ydl_debug_set": (None, [_i, _i]),
    "ydl_debug_attr_sets": (_i, []),
    "ydl_debug_last_kernel": (C.c_char_p, [_i]),
    "ydl_conv_fwd_stats_ws_bytes": (_i64, [_G, _i]),
    "ydl_conv_fwd_grid_m": (_i, [_G, _i]),
    "ydl_conv_fwd_block_m": (_i, [_G, _i]),
    "ydl_conv_fwd": (_i, [_G, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "ydl_conv_fwd_sums": (_i, [_G, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "ydl_conv_dgrad": (_i, [_G, _i, _vp, _vp, _vp, _i, _vp]),
    "ydl_conv_dgrad_bnred_supported": (_i, [_G, _i]),
    "ydl_conv_dgrad_bnred": (_i, [_G, _i, _vp, _vp, _vp, _i, _R, _vp]),
    "ydl_conv_bwd_pw_supported": (_i, [_G, _i]),
    "ydl_conv_bwd_pw": (_i, [_G, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "ydl_conv_wgrad": (_i, [_G, _i, _vp, _vp, _vp, _vp]),
    "ydl_conv_wgrad_ws_bytes": (_i64, [_G, _i]),
    "ydl_conv_wgrad_det": (_i, [_G, _i, _vp, _vp, _vp, _vp, _vp]),
    "ydl_weight_prep": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ydl_weight_prep_batched": (_i, [_i, _vp, _i, _vp]),
    "ydl_wgrad_unpad": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ydl_bn_finalize": (_i, [_vp, _i, _i, _i64, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "ydl_bn_eval_coeffs": (_i, [_i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "ydl_bn_act_fwd": (_i, [_i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i64, _i, _vp]),
    "ydl_bn_act_fwd_sums": (_i, [_i, _vp, _i, _vp, _i, _i64, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _i,
                                 _i64, _i, _i, _vp]),
    "ydl_bn_act_bwd_sums": (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i64, _i, _i,
                                 _vp]),
    "ydl_bn_act_bwd_apply_sums": (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i64, _i, _i,
                                 _vp]),
    "ydl_bn_bwd_ws_bytes": (_i64, [_i64, _i]),
    "ydl_bn_act_bwd": (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i,
                            _vp, _vp, _i, _vp, _i64, _i, _i, _vp]),
    "ydl_sppf_pool_supported": (_i, [_i, _i, _i, _i, _i]),
    "ydl_sppf_pool_fwd": (_i, [_i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ydl_sppf_pool_bwd": (_i, [_i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ydl_maxpool_fwd": (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ydl_maxpool_bwd": (_i, [_i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ydl_resize_fwd": (_i, [_i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _vp]),
    "ydl_resize_bwd": (_i, [_i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _vp]),
    "ydl_resize_acc_sums": (_i, [_i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _vp, _i, _vp]),
    "ydl_copy2d": (_i, [_i, _vp, _i, _vp, _i, _i64, _i, _i, _vp]),
    "ydl_nchw_to_nhwc": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ydl_nchw_to_s2d": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ydl_weight_prep_s2d": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ydl_wgrad_unpack_s2d": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ydl_nhwc_to_nchw": (_i, [_i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "ydl_scale_channels": (_i, [_i, _vp, _i, _vp, _vp, _i, _i, _i64, _i, _vp]),
    "ydl_global_pool_fwd": (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp]),
    "ydl_global_pool_bwd": (_i, [_i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i64, _i, _vp]),
    "ydl_gate_fwd": (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _i, _vp]),
    "ydl_gate_bwd": (_i, [_i, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    "ydl_channel_dot": (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp]),
    "ydl_softmax_fwd": (_i, [_i, _vp, _i, _vp, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _i, _vp]),
    "ydl_softmax_bwd": (_i, [_i, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ydl_seg_loss_ws_floats": (_i64, [_i, _i]),
    "ydl_seg_loss_fwd": (_i, [_vp, _i64, _i64, _i64, _i64, _vp, _i, _i, _vp, _i, _f, _f, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ydl_seg_loss_bwd": (_i, [_vp, _i64, _i64, _i64, _i64, _vp, _i, _i, _vp, _i, _f, _f, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ydl_seg_loss_rep_fwd": (_i, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _i, _f, _f, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ydl_seg_loss_rep_bwd": (_i, [_vp, _i64, _i64, _i64, _i64, _vp, _vp, _i, _f, _f, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ydl_sgd_ema_step": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _f, _f, _f, _f, _f, _i, _f, _vp]),
    "ydl_sgd_ema_step_dev": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _i, _i, _i, _i, _vp]),
    "ydl_sgd_ema_step_multi": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _vp, _i, _vp]),
    "ydl_confusion_matrix": (_i, [_vp, _i64, _i64, _i64, _i64, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ydl_dcnv3_set_border_rule": (None, [_i]),
    "ydl_dcnv3_get_border_rule": (_i, []),
    "ydl_dcnv3_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i, _vp]),
    "ydl_dcnv3_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f,
                           _i, _i, _i, _i, _i, _vp]),
    "ydl_dwconv_fwd": (_i, [_i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ydl_dwconv_dgrad": (_i, [_i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ydl_dwconv_wgrad_ws_bytes": (_i64, [_i, _i]),
    "ydl_dwconv_wgrad": (_i, [_i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ydl_bn_stats_block_m": (_i, []),
    "ydl_bn_stats_ws_bytes": (_i64, [_i64, _i]),
    "ydl_bn_stats": (_i, [_i, _vp, _i, _vp, _i64, _i, _vp]),
    "ydl_channel_sum_ws_bytes": (_i64, [_i]),
    "ydl_channel_sum": (_i, [_i, _vp, _i, _vp, _vp, _i64, _i, _i, _vp]),
    "ydl_group_softmax_fwd": (_i, [_i, _vp, _i, _vp, _i, _i64, _i, _i, _vp]),
    "ydl_group_softmax_bwd": (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _i, _i64, _i, _i, _vp]),
    "ydl_cast_f32": (_i, [_i, _vp, _i, _vp, _i, _i64, _i, _i, _vp]),
    "ydl_reduce_chunks": (_i, [_i, _vp, _vp, _i64, _i, _vp]),
    "ydl_cast_to_f32": (_i, [_i, _vp, _vp, _i64, _i, _vp]),
    "ydl_letterbox_image": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _vp]),
    "ydl_letterbox_mask": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "ydl_fill_zero": (_i, [_vp, _i64, _vp]),
    "ydl_zero2d": (_i, [_i, _vp, _i, _i64, _i, _vp]),
    "ydl_replay_create": (_vp, []),
    "ydl_replay_destroy": (None, [_vp]),
    "ydl_replay_fn_count": (_i, []),
    "ydl_replay_fn_name": (C.c_char_p, [_i]),
    "ydl_replay_add_call": (_i, [_vp, _i, C.POINTER(_i64), _i, _i]),
    "ydl_replay_add_event_record": (_i, [_vp, _i, _i]),
    "ydl_replay_add_event_wait": (_i, [_vp, _i, _i]),
    "ydl_replay_size": (_i, [_vp]),
    "ydl_replay_run": (_i, [_vp, _i, _i, C.POINTER(_vp), _i]),
}

_lib = None


def lib():
    """Load (once) and return the CDLL; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m yolo_dual_amd.build` (hipcc, gfx950). "
                "There is no CPU/PyTorch fallback for the kernels.")
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


_DEBUG_EPOCH = [0]


def debug_set(key: int, val: int) -> None:
    """test-only knobs of the library (ydl.h: ydl_debug_set); bumps an epoch that invalidates cached launch geometry"""
    lib().ydl_debug_set(key, val)
    _DEBUG_EPOCH[0] += 1


def debug_epoch() -> int:
    return _DEBUG_EPOCH[0]


def last_kernel(family: int) -> str:
    """name of the kernel instantiation the last call of an entry family launched (0 fwd, 1 dgrad, 2 wgrad, 3 bn_finalize)"""
    return lib().ydl_debug_last_kernel(family).decode()


class YdlError(RuntimeError):
    pass


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().ydl_last_error()
        raise YdlError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


_PROFILE = None     # list of (name, start_event, end_event, geom-or-None) while profiling is on


def profile_begin() -> None:
    """Bracket every C-ABI launch with events on torch's current stream (the stream the kernels are launched on)."""
    global _PROFILE
    _PROFILE = []


def profile_end():
    """-> list of dicts {name, ms, flops} (flops only for the conv entry points: 2*M*K*Cout, logical channels)."""
    global _PROFILE
    import torch
    rec, _PROFILE = _PROFILE, None
    torch.cuda.synchronize()
    out = []
    for name, e0, e1, g in rec:
        flops, nbytes = 0.0, 0.0
        if isinstance(g, ConvGeom):
            flops = 2.0 * g.N * g.Ho * g.Wo * g.Cout * g.k * g.k * g.Cin
            # algorithmic bytes: input, output and weights once (the weight gradient is f32)
            es = getattr(g, "_es", 2)
            wb = float(g.Cout) * g.k * g.k * g.Cin
            nbytes = (float(g.N) * g.Hi * g.Wi * g.Cin + float(g.N) * g.Ho * g.Wo * g.Cout) * es + wb * (4 if "wgrad" in name else es)
            if getattr(g, "_both", False):
                flops *= 2.0
                nbytes += float(g.N) * g.Hi * g.Wi * g.Cin * es + wb * 4
        elif isinstance(g, float):
            nbytes, g = g, None
        d = {"name": name, "ms": e0.elapsed_time(e1), "flops": flops, "bytes": nbytes}
        if g is not None:
            d["geom"] = [getattr(g, f) for f, _ in ConvGeom._fields_]
            d["kernel"], d["accumulate"] = getattr(g, "_kernel", ""), getattr(g, "_acc", 0)
        out.append(d)
    return out


def _bn_bwd_passes(rmode, act, dres) -> int:
    """tensor passes of one BatchNorm-backward launch, each distinct tensor once: y and dout read, dy written; ReLU also reads the
    stored output (its mask); a residual operand gets its gradient written by the same launch (read-modify-write when a second writer)"""
    n = 3 + (1 if act == ACT_RELU else 0)
    if dres is not None and getattr(dres, "value", dres):
        n += 1 + (1 if (int(rmode) & RES_GRAD_ACCUMULATE) else 0)
    return n


_LAUNCHES = [0]
_RECORDER = None        # yolo_dual_amd.replay.Recorder while a step is being recorded into a launch list


def set_recorder(r) -> None:
    global _RECORDER
    _RECORDER = r


def recorder():
    return _RECORDER


def launch_count() -> int:
    """number of entry-point calls so far (stream-fork bookkeeping: 'has anything been enqueued since ...')"""
    return _LAUNCHES[0]


def call(name: str, *args):
    """Call an int-returning entry point and raise on error."""
    _LAUNCHES[0] += 1
    if _RECORDER is not None:
        _RECORDER.add(name, args)       # the call is recorded AND executed: the recording pass is a real step
    if _PROFILE is None:
        check(getattr(lib(), name)(*args), name)
        return
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(getattr(lib(), name)(*args), name)
    e1.record()
    g = None
    if name in ("ydl_conv_fwd", "ydl_conv_fwd_sums", "ydl_conv_dgrad", "ydl_conv_dgrad_bnred", "ydl_conv_wgrad", "ydl_conv_wgrad_det",
                "ydl_conv_bwd_pw"):
        src = args[0]._obj
        g = ConvGeom(*[getattr(src, f) for f, _ in ConvGeom._fields_])
        g._es = 4 if args[1] == YDL_F32 else 2
        g._both = name == "ydl_conv_bwd_pw"          # input AND weight gradient: twice the FLOPs, x + dy + dx + dw bytes
        g._kernel = last_kernel(1 if ("dgrad" in name or "bwd_pw" in name) else 2 if "wgrad" in name else 0)
        g._acc = int(args[5]) if "dgrad" in name else int(args[6]) if name.startswith("ydl_conv_fwd") else int(args[7]) if g._both else 0
    elif name == "ydl_bn_act_fwd":          # algorithmic bytes: y (+ residual) read once, out written once
        es = 4 if args[0] == YDL_F32 else 2
        g = float(args[11]) * args[12] * es * (2 + (1 if args[7] else 0))
    elif name == "ydl_bn_act_bwd":          # y and dout read once, dy written once (the two-phase kernel reads them twice)
        es = 4 if args[0] == YDL_F32 else 2
        g = float(args[22]) * args[24] * es * _bn_bwd_passes(args[12], args[13], args[16])
    elif name == "ydl_bn_act_fwd_sums":
        es = 4 if args[0] == YDL_F32 else 2
        g = float(args[23]) * args[25] * es * (2 + (1 if args[19] else 0))
    elif name == "ydl_bn_act_bwd_sums":
        es = 4 if args[0] == YDL_F32 else 2
        g = float(args[21]) * args[23] * es * _bn_bwd_passes(args[11], args[12], args[15])
    elif name in ("ydl_dcnv3_fwd", "ydl_dcnv3_bwd"):
        # algorithmic bytes: input, offsets, masks (and grad_output) read once; output / the three f32 gradients written once
        es = 4 if args[0] == YDL_F32 else 2
        o = 0 if name == "ydl_dcnv3_fwd" else 3
        kh, kw, G, Gc = args[5 + o], args[6 + o], args[13 + o], args[14 + o]
        N, H, W, Ho, Wo = args[16 + o:21 + o]
        C, pts = G * Gc, G * kh * kw * 3
        g = float(es) * (N * H * W * C + N * Ho * Wo * (C + pts))
        if o:
            g += 4.0 * (N * H * W * C + N * Ho * Wo * pts)
    _PROFILE.append((name, e0, e1, g))
