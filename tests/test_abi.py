"""CPU: the C-ABI library loads, exports every symbol include/ydl.h declares, and the ctypes table mirrors the header
(argument counts).  No compute calls (there is no GPU in the build container)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "ydl.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int64_t|int|void|const char\*|ydl_replay\*)\s+(ydl_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[m.group(1)] = n
    return out


def test_library_exports_every_declared_symbol():
    from yolo_dual_amd import _lib
    from yolo_dual_amd.build import build
    build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    decl = _header_functions()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in ydl.h but not exported"


def test_ctypes_table_matches_header():
    from yolo_dual_amd import _lib
    decl = _header_functions()
    assert set(decl) == set(_lib.SIGNATURES), (set(decl) ^ set(_lib.SIGNATURES))
    for name, n in decl.items():
        assert len(_lib.SIGNATURES[name][1]) == n, (name, n, len(_lib.SIGNATURES[name][1]))


def test_error_path_without_gpu_call():
    """argument validation happens on the host before any launch: a bad geometry returns an error string"""
    from yolo_dual_amd import _lib as L
    g = L.ConvGeom(1, 8, 8, 8, 9, 9, 8, 3, 1, 1, 8, 8, 0)     # wrong Ho/Wo
    rc = L.lib().ydl_conv_fwd(ctypes.byref(g), 0, None, None, None, None, 0, None)
    assert rc != 0
    assert b"output size" in L.lib().ydl_last_error()
    assert L.lib().ydl_version() >= 1


def test_cpu_tensor_is_rejected_loudly():
    import pytest
    import torch
    import yolo_dual_amd as ydl
    m = ydl.Conv(8, 8, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 8, 4, 4))
    with pytest.raises(RuntimeError, match="GPU only"):
        ydl.SegmentationLoss(12)(torch.zeros(1, 12, 4, 4), torch.zeros(1, 4, 4, dtype=torch.long))


def test_replay_dispatch_table_is_current():
    """csrc/replay_table.inc is generated from ydl.h (tools/gen_replay.py): the committed file equals a fresh generation, and every
    stream-taking entry point of the ctypes table is recordable"""
    import importlib.util
    from yolo_dual_amd import _lib as L
    spec = importlib.util.spec_from_file_location("gen_replay", os.path.join(ROOT, "tools", "gen_replay.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert open(gen.OUT).read() == gen.generate(), "stale replay_table.inc: run python tools/gen_replay.py"
    lib = L.lib()
    names = {lib.ydl_replay_fn_name(i).decode() for i in range(lib.ydl_replay_fn_count())}
    launchers = {n for n, (res, args) in L.SIGNATURES.items()
                 if res is L._i and args and args[-1] is L._vp and not n.startswith("ydl_replay_")
                 and n not in ("ydl_debug_set",)}
    assert launchers <= names, launchers - names
    assert "ydl_conv_fwd" in names and "ydl_fill_zero" in names and "ydl_sgd_ema_step_dev" in names


def test_launch_list_recording_on_the_host():
    """the Recorder mirrors calls into the C-side list without a GPU: argument conversion (geometry copied by value, floats,
    pointers, stream slots), event edges, range checks of ydl_replay_run; an empty range runs nothing"""
    import pytest
    from yolo_dual_amd import _lib as L
    from yolo_dual_amd.replay import Recorder
    rec = Recorder()
    g = L.ConvGeom(1, 8, 8, 8, 8, 8, 8, 3, 1, 1, 8, 8, 0)
    rec.add("ydl_conv_fwd", (ctypes.byref(g), 1, ctypes.c_void_p(4096), ctypes.c_void_p(8192), ctypes.c_void_p(12288), None, 0,
                             ctypes.c_void_p(0)))
    g.N = 99                                     # the recorded copy is independent of the caller's struct
    rec.add("ydl_bn_finalize", (ctypes.c_void_p(16), 4, 128, 1 << 33, 8, None, None, 1e-5, ctypes.c_float(0.1), None, None, None, None,
                                None, None, 1, ctypes.c_void_p(0x1234)))
    rec.edge(0, 0x1234)
    assert rec.size() == 4 and rec.calls == 2 and rec.handles == [0, 0x1234]
    with pytest.raises(RuntimeError, match="not a recordable"):
        rec.add("ydl_conv_fwd_grid_m", (ctypes.byref(g), 1))
    with pytest.raises(TypeError):
        rec.add("ydl_copy2d", (1, None))
    rec.run(0, 0)                                # nothing to issue: no device needed
    with pytest.raises(L.YdlError, match="out of bounds"):
        rec.run(0, 99)
    rec.close()
