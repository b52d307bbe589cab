"""Host check of the device arithmetic header (image_matching_amd/csrc/devmath.h) against unsigned __int128 %: single-word
Barrett, Shoup, reduce64, double-word Barrett and the four-product lazy-sum reduction, on the engine's moduli and at range edges."""
import os
import subprocess

from conftest import ROOT


def test_devmath_against_int128(tmp_path):
    exe = tmp_path / "devmath_check"
    src = os.path.join(ROOT, "tests", "csrc", "devmath_check.cpp")
    inc = os.path.join(ROOT, "image_matching_amd", "csrc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", inc, src, "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "devmath ok" in out.stdout, out.stdout + out.stderr


def test_database_address_map_is_a_bijection(tmp_path):
    """image_matching_amd/csrc/db_layout.h on the host: both resident layouts tile the allocation exactly (no overlap, no hole), and in
    the group-sequential one a loop-B workgroup's bytes are one contiguous run — the property the layout exists for."""
    exe = tmp_path / "db_layout_check"
    src = os.path.join(ROOT, "tests", "csrc", "db_layout_check.cpp")
    inc = os.path.join(ROOT, "image_matching_amd", "csrc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", inc, src, "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "db layout ok" in out.stdout, out.stdout + out.stderr
