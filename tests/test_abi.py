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
    for m in re.finditer(r"\b(?:int64_t|int|void|const char\*)\s+(ydl_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
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
