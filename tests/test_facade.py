"""The C++ facade (include/mc_marching.hpp) compiled as a reference-style caller."""
import subprocess
import sys

import pytest

from conftest import ROOT

EXE = ROOT / "tests" / "native" / "facade_demo"


def build_demo(mc):
    src = ROOT / "tests" / "native" / "facade_demo.cpp"
    if not EXE.exists() or EXE.stat().st_mtime < max(src.stat().st_mtime, (ROOT / "include" / "mc_marching.hpp").stat().st_mtime,
                                                      mc.LIB_PATH.stat().st_mtime):
        subprocess.run(["g++", "-std=c++17", "-O1", f"-I{ROOT / 'include'}", str(src), "-o", str(EXE), f"-L{mc.LIB_PATH.parent}",
                        "-lmc_hip", f"-Wl,-rpath,{mc.LIB_PATH.parent}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return EXE


def test_facade_compiles_and_links(mc):
    """A plain g++ caller needs only the two headers and libmc_hip.so (no HIP headers, no torch)."""
    assert build_demo(mc).exists()


def test_facade_without_gpu_reports_error(mc):
    if mc.device_count() > 0:
        pytest.skip("a GPU is visible here")
    r = subprocess.run([str(build_demo(mc))], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("eq,n,iso,tris,fnv", [
    ("x^2+y^2+z^2-1", 32, 0.0, 9548, "4598d5da3647dd7b"),      # SURVEY.md section 4 reference fingerprints
    ("x+y", 32, 0.0, 4290, "3081dde6768e362b"),
    ("(x^2)^2+(y^2)^2+(z^2)^2-(x^2+y^2+z^2)", 32, -0.4, 16912, "61136007ac533813"),
])
def test_facade_matches_reference_fingerprints(mc, eq, n, iso, tris, fnv):
    r = subprocess.run([str(build_demo(mc)), eq, str(n), str(iso)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"cells_per_axis={n + 1} tris={tris} verts={3 * tris} fnv_soup={fnv}" in r.stdout
