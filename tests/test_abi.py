"""The C-ABI shared library loads without a GPU and exports exactly the symbols include/hyslam_amd.h declares.
No compute entry point is called here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hyslam_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from hyslam_amd import _native as N
    lib = N.lib()
    decl = declared_symbols()
    assert len(decl) >= 20
    for sym in decl:
        assert hasattr(lib, sym), "libhyslam_amd.so does not export " + sym
    assert sorted(N.EXPORTS) == decl, "python binding list and header disagree"


def test_struct_layouts_match_header():
    from hyslam_amd import _native as N
    assert C.sizeof(N.OrbParams) == 44 and C.sizeof(N.StereoParams) == 24 and N.KP_DTYPE.itemsize == 24
    p = N.OrbParams()
    N.lib().hs_orb_default_params(C.byref(p))       # pure host code: ORBFactory defaults (ORBFactory.cpp:13-25)
    assert (p.nfeatures, p.nlevels, p.cell_px, p.ini_th_fast, p.min_th_fast, p.fast_threshold) == (1000, 8, 30, 20, 4, 20)
    assert abs(p.scale_factor - 1.2) < 1e-6 and list(p.blur_taps) == [18, 34, 49, 55, 49, 34, 18]


def test_status_strings_and_null_handles():
    from hyslam_amd import _native as N
    lib = N.lib()
    assert lib.hs_status_string(0) == b"ok" and b"capacity" in lib.hs_status_string(3)
    assert b"gfx950" in lib.hs_version()
    assert lib.hs_orb_get_levels(None) == 0 and lib.hs_orb_max_keypoints(None) == 0
    lib.hs_orb_destroy(None)
    assert lib.hs_orb_create(None, 0, None) == N.HS_ERR_INVALID
    h = C.c_void_p()
    bad = N.OrbParams()
    lib.hs_orb_default_params(C.byref(bad))
    bad.nlevels = 0
    assert lib.hs_orb_create(C.byref(bad), 0, C.byref(h)) == N.HS_ERR_INVALID and not h.value


def test_no_silent_cpu_fallback(monkeypatch):
    """With the extension missing the product must raise, never compute on the CPU."""
    from hyslam_amd import _native as N
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", os.path.join(ROOT, "hyslam_amd", "does_not_exist.so"))
    with pytest.raises(ImportError):
        N.lib()


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hyslam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "hs_oracle" not in src and "import oracle" not in src and "oracle/" not in src, os.path.join(dirpath, f)
