"""The describe kernel's sin/cos (hyslam_amd/csrc/lean_sincos.h: IEEE double operations only, so host and device agree bit for bit) against
libm, rounded to float as ORBFinder.cpp:95-98 does, for EVERY float rotation angle in [0, 6.5] rad — 1.09e9 values, split over the cores."""
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lean_sincos_equals_libm_for_every_float_angle(tmp_path):
    exe = str(tmp_path / "sincos_exhaustive")
    src = os.path.join(ROOT, "tests", "cpp", "sincos_exhaustive.cpp")
    base = ["g++", "-O2", "-ffp-contract=off", "-o", exe, src, "-lm"]
    if subprocess.run(base[:2] + ["-mfma"] + base[2:], capture_output=True).returncode != 0:
        subprocess.run(base, check=True)
    top = struct.unpack("<I", struct.pack("<f", 6.5))[0]
    n = max(1, min(8, os.cpu_count() or 1))
    edges = [top * i // n for i in range(n + 1)]
    procs = [subprocess.Popen([exe, hex(edges[i]), hex(edges[i + 1])], stdout=subprocess.PIPE, text=True) for i in range(n)]
    outs = [p.communicate()[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert sum(int(o.split()[0]) for o in outs) == top
