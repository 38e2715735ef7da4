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


CALLER = ROOT / "tests" / "native" / "reference_caller"


def build_caller(mc):
    """tests/native/reference_caller.cpp: Source/main.cpp:11-14 + marching_test_drawer.h:7-15, its #include lines as they
    are in the reference, compiled against include/compat."""
    src = ROOT / "tests" / "native" / "reference_caller.cpp"
    if not CALLER.exists() or CALLER.stat().st_mtime < max(src.stat().st_mtime, (ROOT / "include" / "mc_marching.hpp").stat().st_mtime,
                                                            mc.LIB_PATH.stat().st_mtime):
        subprocess.run(["g++", "-std=c++14", "-O1", f"-I{ROOT / 'include' / 'compat'}", str(src), "-o", str(CALLER),
                        f"-L{mc.LIB_PATH.parent}", "-lmc_hip", f"-Wl,-rpath,{mc.LIB_PATH.parent}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return CALLER


def test_reference_style_caller_compiles_with_only_the_include_path_changed(mc):
    assert build_caller(mc).exists()


@pytest.mark.gpu
@pytest.mark.parametrize("eq,verts,tris", [("x^2+y^2+z^2-1", 4758, 9548), ("x+y", 1122, 4290)])   # SURVEY.md section 4 counts
def test_reference_style_caller_gets_the_reference_mesh(mc, orc, eq, verts, tris):
    """Evaluator() / Marching() without arguments, evaluate(x, y, z), recalculate(), get_poly_data(), CalculateNormal(pData):
    the welded mesh equals the oracle's replay of the reference's std::set welding, hash for hash."""
    import numpy as np
    r = subprocess.run([str(build_caller(mc)), eq], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    ref = orc.march_indexed(eq, float(np.float32(2.0) / np.float32(32)), pow_mode=orc.POW_EXACT)
    assert (ref.n_verts, ref.n_tris) == (verts, tris)
    want = (f"verts={verts} tris={tris} fnv_vertex_list={orc.fnv1a(ref.vertices.tobytes()):016x} "
            f"fnv_tri_list={orc.fnv1a(ref.tris.tobytes()):016x} normals={verts} ")
    assert want in r.stdout, (r.stdout, want)


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


# indexed vertex / triangle counts of the unmodified reference (std::set welding)
@pytest.mark.gpu
@pytest.mark.parametrize("eq,n,iso,scale,step,verts,tris", [
    ("x+y", 32, 0.0, 1.0, 0.0, 1122, 4290),
    ("x^2+y^2+z^2-1", 32, 0.0, 1.0, 0.0, 4758, 9548),
    ("x^2+y^2+z^2-1", 64, 0.0, 1.0, 0.0, 19230, 38492),
    ("x^2*y^2+x^2*z^2+z^2*y^2+x*y*z", 32, 0.0, 1.0, 0.0, 1188, 4612),
    ("(x^2+y^2+z^2+(1/3)^2-(1/5)^2)^2-4*((1/2)*x-(2.36/6)*(1/5))^2-4*(1/3)^2*y^2", 32, 0.0, 1.0, 0.0, 1237, 2436),
    ("(x^2+y^2-(1/16))^2+(y^2+z^2-(1/16))^2+(z^2+x^2-(1/16))^2-8*(x^2+y^2+z^2-(1/4))^2", 32, 0.0, 1.0, 0.0, 2820, 5632),
    ("(x^2)^2+(y^2)^2+(z^2)^2-(x^2+y^2+z^2)", 32, -0.4, 1.0, 0.0, 8628, 16912),
    ("(x^2+y^2-1)^2 + (x^2+z^2-1)^2 + (z^2+y^2-1)^2 - 0.5", 10, 0.0, 1.1, 0.2, 648, 1312),
    ("(x-0.1)*(y-0.07)-0.001", 4, 0.0, 1.0, 0.0, 72, 100),
    ("(x-0.1)*(y-0.07)*(z-0.13)-0.0001", 4, 0.0, 1.0, 0.0, 108, 148),
])
def test_facade_indexed_mesh_matches_reference_counts(mc, eq, n, iso, scale, step, verts, tris):
    """set_indexed(true), the facade's default: the reference's vertex welding (marching.cpp:599-654) done on the GPU."""
    r = subprocess.run([str(build_demo(mc)), eq, str(n), str(iso), "indexed", str(scale), str(step)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"tris={tris} verts={verts} " in r.stdout, r.stdout


@pytest.mark.gpu
def test_facade_ply_round_trip(mc, tmp_path):
    """save_poly_to_file / load_poly_from_file use the reference's ASCII PLY (marching.cpp:665-854)."""
    ply = tmp_path / "mesh.ply"
    r = subprocess.run([str(build_demo(mc)), "x^2+y^2+z^2-1", "16", "0", "indexed", "1", "0", str(ply)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ply_roundtrip=ok" in r.stdout and float(r.stdout.split("maxd=")[1].split()[0]) < 1e-6
    head = ply.read_text().splitlines()[:9]
    assert head[0] == "ply" and head[1] == "format ascii 1.0" and head[2].startswith("element vertex ")
    assert head[3:6] == ["property float x", "property float y", "property float z"]
    assert head[6].startswith("element face ") and head[6].endswith(" ")      # the reference's trailing blank
    assert head[7] == "property list uchar int vertex_indices" and head[8] == "end_header"


@pytest.mark.gpu
def test_facade_constraint(mc):
    """set_constraint0("x", ">", -0.5) + use_constraint0(true), as the reference's developer viewer does;
    expected count and fingerprint from the oracle (tests/test_oracle_pins.py pins its constraint semantics)."""
    import os
    r = subprocess.run([str(build_demo(mc)), "x^2+y^2+z^2-1", "32", "0"], capture_output=True, text=True,
                       env=dict(os.environ, MC_DEMO_CONSTRAINT="x > -0.5"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "tris=6772 " in r.stdout and "fnv_soup=c18a9ae00b655374" in r.stdout, r.stdout


@pytest.mark.gpu
def test_facade_indexed_normals_are_calculate_normal(mc, tmp_path):
    """set_indexed(true): normal_list = the reference drawer's CalculateNormal (normal.h:3-41: area-weighted sum of
    face normals per welded vertex, normalised), recomputed here in float32 from the saved PLY."""
    import os
    import numpy as np
    ply, nrm = tmp_path / "m.ply", tmp_path / "n.txt"
    r = subprocess.run([str(build_demo(mc)), "x^2+y^2+z^2-1", "24", "0", "indexed", "1", "0", str(ply)], capture_output=True, text=True,
                       env=dict(os.environ, MC_DEMO_NORMALS=str(nrm)))
    assert r.returncode == 0, r.stdout + r.stderr
    lines = ply.read_text().splitlines()
    nv, nf = int(lines[2].split()[2]), int(lines[6].split()[2])
    v = np.array([l.split() for l in lines[9:9 + nv]], dtype=np.float32)
    f = np.array([l.split()[1:] for l in lines[9 + nv:9 + nv + nf]], dtype=np.int64)
    got = np.loadtxt(nrm, dtype=np.float32).reshape(-1, 3)
    assert got.shape == (nv, 3)
    acc = np.zeros((nv, 3), np.float32)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    fn = np.cross((b - a).astype(np.float32), (c - a).astype(np.float32)).astype(np.float32)
    for k in range(3):
        np.add.at(acc, f[:, k], fn)
    want = acc / np.linalg.norm(acc, axis=1, keepdims=True)
    # the PLY stores %f (6 decimals), so the recomputation sees rounded vertices: agreement to ~1e-4, and outward
    assert np.nanmax(np.abs(got - want)) < 2e-3
    assert (np.sum(got * v, axis=1) > 0.9).all()          # sphere: normals point outwards, like the winding
    assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("eq,n,iso,tris,fnv_codes,fnv_soup", [
    ("x^2+y^2+z^2-1", 16, 0.0, None, None, None),                                     # against the sweep itself
    ("(x-0.1)*(y-0.07)*(z-0.13)-0.0001", 4, 0.0, None, "3fe2175923bab521", "4b4bd108c366fe8f"),
    ("(x-0.1)*(y+0.07)*(z-0.13)-0.0005", 4, 0.0, None, "38924c48d8f5cae0", "ab531bb809938f33"),   # (both ambiguity outcomes)
])
def test_facade_step_trace_of_every_cell_rebuilds_the_sweep(mc, eq, n, iso, tris, fnv_codes, fnv_soup):
    """Marching::step_at (the reference's step-by-step Step_Data for one cell, marching.cpp:456-595) called for every
    cell in sweep order: the concatenated triangles are the sweep's soup and the codes its code volume, bit for bit --
    for the two ambiguity rows that is the fingerprint SURVEY.md recorded from the unmodified reference."""
    import os
    r = subprocess.run([str(build_demo(mc)), eq, str(n), str(iso)], capture_output=True, text=True, env=dict(os.environ, MC_DEMO_STEPS="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    sweep = [l for l in r.stdout.splitlines() if l.startswith("cells_per_axis")][0]
    steps = [l for l in r.stdout.splitlines() if l.startswith("steps:")][0]
    assert steps.split("tris=")[1].split()[0] == sweep.split("tris=")[1].split()[0]
    assert steps.split("fnv_soup=")[1].split()[0] == sweep.split("fnv_soup=")[1].split()[0]
    if fnv_soup:
        assert f"fnv_codes={fnv_codes}" in steps and f"fnv_soup={fnv_soup}" in steps


@pytest.mark.gpu
def test_facade_seed_mode(mc):
    """seed_mode(true) + set_seed(1,0,0) on the unit sphere: the count of the oracle's restatement of the reference's walk
    (tests/test_oracle_pins.py); seed (0,0,0) sits in a cell without a crossing and yields nothing."""
    import os
    for seed, tris in (("1 0 0", 9537), ("0 0 0", 0)):
        r = subprocess.run([str(build_demo(mc)), "x^2+y^2+z^2-1", "32", "0"], capture_output=True, text=True,
                           env=dict(os.environ, MC_DEMO_SEED=seed))
        assert r.returncode == 0, r.stdout + r.stderr
        assert f" tris={tris} " in r.stdout, r.stdout
    # the facade's default, set_indexed(true): get_poly_data() is the welded mesh in seed mode too (marching.cpp:310-331)
    r = subprocess.run([str(build_demo(mc)), "x^2+y^2+z^2-1", "32", "0", "indexed"], capture_output=True, text=True,
                       env=dict(os.environ, MC_DEMO_SEED="1 0 0"))
    assert r.returncode == 0, r.stdout + r.stderr
    verts = int(r.stdout.split("verts=")[1].split()[0])
    assert " tris=9537 " in r.stdout and 4000 < verts < 5000, r.stdout


@pytest.mark.gpu
def test_facade_objects_on_one_context_keep_their_own_state(mc):
    """Two Marching() objects share the process-wide context; constraints, seed mode and the seed are per object as in the
    reference (marching.h:58-69, :130-157): one object's constraint must not clip the other's sweep, its seed mode must not
    change the other's mesh, and a destroyed object leaves nothing behind."""
    import os
    r = subprocess.run([str(build_demo(mc)), "x^2+y^2+z^2-1", "32", "0"], capture_output=True, text=True,
                       env=dict(os.environ, MC_DEMO_TWO_OBJECTS="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "two_objects=ok plain=9548 constrained=6772" in r.stdout, r.stdout   # 6772: test_facade_constraint's oracle count
