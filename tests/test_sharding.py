"""Z-slab sharding across ranks (one process per GPU).  CPU: world_size-2 gloo run of the host
logic with the oracle standing in for the device (test infrastructure); GPU: two ranks sharing
the one GPU of the box run the real bench path with a gloo rendezvous."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import EQ, ROOT

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/oracle")
import mc_amd, pyoracle as orc
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
eq, step = "x^2+y^2+z^2-1", float(np.float32(2.0) / np.float32(24))
n1 = mc_amd.cells_per_axis(step)
zb, ze = mc_amd.shard_layers(n1, world, rank)
m = orc.march(eq, step, pow_mode=orc.POW_EXACT, want=orc.WANT_CODES | orc.WANT_SOUP, z_begin=zb, z_end=ze, nthreads=2)
counts = torch.zeros(world, dtype=torch.int64)
dist.all_gather_into_tensor(counts, torch.tensor([m.n_tris], dtype=torch.int64))
offsets, total = mc_amd.exclusive_offsets(counts.tolist())
# every rank writes its slab at its offset of a shared result; rank 0 checks against the unsharded sweep
soup = torch.zeros(total * 9, dtype=torch.float32)
soup[offsets[rank] * 9:(offsets[rank] + m.n_tris) * 9] = torch.from_numpy(m.soup.reshape(-1))
dist.all_reduce(soup)   # disjoint ranges: a sum is a concatenation
if rank == 0:
    whole = orc.march(eq, step, pow_mode=orc.POW_EXACT, want=orc.WANT_SOUP, nthreads=2)
    assert total == whole.n_tris, (total, whole.n_tris)
    assert np.array_equal(soup.numpy().view(np.uint32), whole.soup.reshape(-1).view(np.uint32))
    print("SHARD_OK", total, offsets)
dist.destroy_process_group()
'''


WORKER_SEAM = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/oracle"); sys.path.insert(0, sys.argv[1] + "/tests")
import mc_amd, pyoracle as orc
from test_weld_model import seam_slab_by_keys
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
eq, step = "x^2+y^2+z^2-1", float(np.float32(2.0) / np.float32(16))
n1 = mc_amd.cells_per_axis(step)
zb, ze = mc_amd.shard_layers(n1, world, rank)
# what the rank's GPU does with MC_FLAG_INDEXED | MC_FLAG_SEAM, stated by the CPU model of the rule (tests/test_weld_model.py)
v, t = seam_slab_by_keys(eq, step, 0.0, 1.0, (), zb, ze, n1)
counts = torch.zeros(2 * world, dtype=torch.int64)
dist.all_gather_into_tensor(counts, torch.tensor([len(v), len(t)], dtype=torch.int64))   # the one exchange: the counts
voff, vtot = mc_amd.exclusive_offsets(counts[0::2].tolist())
toff, ttot = mc_amd.exclusive_offsets(counts[1::2].tolist())
t = t + voff[rank]                                              # mc_index_rebase(ctx, offset)
V = torch.zeros(vtot * 3, dtype=torch.float32); T = torch.zeros(ttot * 3, dtype=torch.int64)
V[voff[rank] * 3:(voff[rank] + len(v)) * 3] = torch.from_numpy(np.ascontiguousarray(v).reshape(-1))
T[toff[rank] * 3:(toff[rank] + len(t)) * 3] = torch.from_numpy(np.ascontiguousarray(t).reshape(-1))
dist.all_reduce(V); dist.all_reduce(T)                          # disjoint ranges: a sum is a concatenation
if rank == 0:
    ref = orc.march_indexed(eq, step, pow_mode=orc.POW_EXACT)   # the reference's std::set welding of the WHOLE grid
    assert (vtot, ttot) == (ref.n_verts, ref.n_tris), (vtot, ttot, ref.n_verts, ref.n_tris)
    assert np.array_equal(V.numpy().view(np.uint32), ref.vertices.reshape(-1).view(np.uint32))
    assert np.array_equal(T.numpy().astype(np.uint32), ref.tris.reshape(-1))
    print("SEAM_OK", vtot, ttot, voff)
dist.destroy_process_group()
'''


def test_two_rank_gloo_one_poly_data_from_two_slabs(tmp_path):
    """world_size 2 on CPU: each rank welds its slab as a part of the whole grid (the MC_FLAG_SEAM rule, CPU model), the
    ranks all-gather their vertex / triangle counts, re-base their tri_list by the offset (mc_index_rebase) and the
    concatenation is the reference's Poly_Data of the whole grid, bit for bit."""
    script = tmp_path / "worker_seam.py"
    script.write_text(WORKER_SEAM)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29537", str(script), str(ROOT)], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "SEAM_OK" in r.stdout


def test_shard_layers_cover_and_order(mc):
    for n in (1, 5, 33, 257, 1025, 2001):
        for world in (1, 2, 3, 4, 8):
            ranges = [mc.shard_layers(n, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert mc.exclusive_offsets([3, 0, 7]) == ([0, 3, 3], 10)


def test_two_rank_gloo_sharded_sweep(tmp_path):
    """world_size 2 on CPU: slabs + all-gathered counts + offsets reproduce the unsharded order."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29533", str(script), str(ROOT)], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "SHARD_OK" in r.stdout


def test_bench_gpus_n_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus N`, started plainly (the way the driver starts it), must become the launcher: N rank
    processes through torch.distributed.run with the same arguments, before this process touches a GPU.  Checked on CPU by
    putting a stand-in `torch.distributed.run` in front of the real one that records how it was called."""
    fake = tmp_path / "torch" / "distributed"
    fake.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("import os, sys\nopen(os.environ['FAKE_TORCH_LOG'], 'a').write(sys.argv[0] + '\\n')\n")
    (fake / "__init__.py").write_text("")
    (fake / "run.py").write_text("import json, os, sys\nprint('SPAWNED ' + json.dumps(sys.argv[1:]))\nsys.exit(7)\n")
    log = tmp_path / "imports.log"
    env = dict(os.environ, PYTHONPATH=str(tmp_path), FAKE_TORCH_LOG=str(log))
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1"], capture_output=True,
                       text=True, env=env, timeout=120)
    assert r.returncode == 7, r.stdout + r.stderr            # the children's exit code is the launcher's
    assert str(ROOT / "bench.py") not in log.read_text().splitlines()   # the launcher itself never imported torch
    line = [l for l in r.stdout.splitlines() if l.startswith("SPAWNED ")][0]
    argv = json.loads(line[len("SPAWNED "):])
    assert "--nproc-per-node=4" in argv and "--nnodes=1" in argv and argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    i = argv.index(str(ROOT / "bench.py"))
    assert argv[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """bench.py's N>1 path (slabs, count all-gather, max-over-ranks timing) with 2 ranks sharing GPU 0, started the way
    the driver starts it: plainly, `python bench.py --gpus 2` (bench.py spawns its ranks itself)."""
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_SINGLE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup",
                        "1", "--grid-res", "256", "--halo-check"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["triangles"] == 617180 and d["config"]["cells"] == 257 ** 3
    assert d["scaling"] == "strong" and d["value"] > 0
    b = d["config"]["z_bounds"]                       # count-balanced slabs: still a partition of all 257 layers
    assert b[0] == 0 and b[-1] == 257 and 0 < b[1] < 257
    # the halo plane a sampled-field design would exchange equals the plane each rank recomputes
    assert d["halo"]["recomputed_plane_identical_to_exchanged"] is True


@pytest.mark.gpu
def test_bench_rccl_path_with_one_rank():
    """The `nccl` (= RCCL) branch of bench.py on the one GPU there is: world_size 1 forced through the distributed path --
    init_process_group with device_id, the stream-ordered copy of the device-side counts, the CUDA all_gather_into_tensor,
    barrier and the max / sum reductions all execute."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--grid-res", "256",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["triangles"] == 617180
    assert d["config"]["count_exchange"].startswith("library: mc_comm_gather_async"), d["config"]["count_exchange"]
    # ... and the torch.distributed form of the same exchange, which bench.py falls back to when the library's cannot be set up
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2", "--grid-res", "256",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=dict(env, BENCH_EXCHANGE="torch", MASTER_PORT="29543"), timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["config"]["triangles"] == 617180 and d["config"]["count_exchange"] == "rccl all_gather_into_tensor, device-side counts"


def test_rebalance_layers_equalises_a_known_cost_profile():
    """Count-balanced Z repartition (host logic): with a per-layer cost like the sphere's undecided rows (~ sqrt(1 - z^2))
    plus a constant, three rounds of 'measure slabs, re-cut' bring the slab costs within a few percent of each other."""
    import numpy as np
    import mc_amd
    n1, world = 1025, 8
    z = (np.arange(n1) + 0.5) / n1 * 2 - 1
    w = 0.3 + np.sqrt(np.clip(1 - z * z, 0, None))                  # cost per layer
    bounds = [mc_amd.shard_layers(n1, world, r)[0] for r in range(world)] + [n1]
    spread0 = None
    for _ in range(3):
        costs = [float(w[bounds[i]:bounds[i + 1]].sum()) for i in range(world)]
        spread0 = spread0 or max(costs) / (sum(costs) / world)
        bounds = mc_amd.rebalance_layers(bounds, costs)
        assert bounds[0] == 0 and bounds[-1] == n1 and all(b > a for a, b in zip(bounds, bounds[1:]))
    costs = [float(w[bounds[i]:bounds[i + 1]].sum()) for i in range(world)]
    assert spread0 > 1.15 and max(costs) / (sum(costs) / world) < 1.04
    # degenerate inputs: nothing to balance, or fewer layers than ranks
    assert mc_amd.rebalance_layers([0, 5, 10], [0.0, 0.0]) == [0, 5, 10]
    assert mc_amd.rebalance_layers([0, 1, 2, 3], [5.0, 1.0, 1.0]) == [0, 1, 2, 3]


@pytest.mark.gpu
def test_bench_in_flight_sweeps_produce_the_same_counts():
    """bench.py keeps several independent sweeps in flight (step k on context k % D).  Every frame of the iso sweep must
    still be produced and counted: the total over the frames is the same with 1, 2 and 3 frames in flight; and the
    headline mode reports the same triangles."""
    totals, tris = set(), set()
    for d in ("1", "2", "3"):
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--mode", "isosweep", "--grid-res", "128", "--steps", "13", "--warmup", "2",
                            "--in-flight", d], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert j["config"]["in_flight"] == int(d) and j["config"]["frames"] == 13
        totals.add(j["config"]["triangles_total"])
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--grid-res", "192", "--steps", "7", "--warmup", "2", "--in-flight", d,
                            "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert j["config"]["in_flight"] == int(d)
        tris.add(j["config"]["triangles"])
    assert len(totals) == 1 and len(tris) == 1


# ---- the multi-device entry points of the C ABI (mc_march_sharded, mc_comm_*: include/mc_hip.h) ----------------------
SHARDED_CALLER = ROOT / "tests" / "native" / "sharded_caller"


def build_sharded_caller(mc):
    """tests/native/sharded_caller.cpp: a reference-style caller (the reference's #include lines, include/compat) that gives
    its Marching object a device list."""
    src = ROOT / "tests" / "native" / "sharded_caller.cpp"
    if not SHARDED_CALLER.exists() or SHARDED_CALLER.stat().st_mtime < max(src.stat().st_mtime, (ROOT / "include" / "mc_marching.hpp").stat().st_mtime,
                                                                            mc.LIB_PATH.stat().st_mtime):
        subprocess.run(["g++", "-std=c++14", "-O1", "-g", "-rdynamic", f"-I{ROOT / 'include' / 'compat'}", str(src), "-o", str(SHARDED_CALLER),
                        f"-L{mc.LIB_PATH.parent}", "-lmc_hip", f"-Wl,-rpath,{mc.LIB_PATH.parent}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return SHARDED_CALLER


def test_sharded_caller_compiles_against_the_facade(mc):
    assert build_sharded_caller(mc).exists()


def test_c_shard_layers_is_the_python_rule(mc):
    """mc_shard_layers (C ABI, what mc_march_sharded / mc_march_rank cut by) == shard_layers (the host logic the gloo tests run)."""
    import ctypes as C
    L = mc.lib()
    for n in (0, 1, 5, 33, 257, 1025, 2001):
        for world in (1, 2, 3, 4, 8):
            for r in range(world):
                a, b = C.c_int(), C.c_int()
                L.mc_shard_layers(n, world, r, C.byref(a), C.byref(b))
                assert (a.value, b.value) == mc.shard_layers(n, world, r)


def test_sharded_entry_points_refuse_bad_arguments_without_a_gpu(mc):
    """Argument checks come before any device work: an empty list, a null context, descending bounds."""
    import ctypes as C
    L = mc.lib()
    p = mc.McParams()
    p.equation, p.step, p.scale, p.z_end = b"x+y", 0.25, (C.c_float * 3)(1, 1, 1), -1
    res, sh = (mc.McResult * 2)(), (mc.McShard * 2)()
    arr = (C.c_void_p * 2)(None, None)
    assert L.mc_march_sharded(arr, 0, C.byref(p), None, res, sh) == mc.MC_ERR_ARG
    assert L.mc_march_sharded(arr, 2, C.byref(p), None, res, sh) == mc.MC_ERR_ARG and b"null" in L.mc_last_error()
    assert L.mc_copy_sharded_vertices(arr, 2, None, 0) == mc.MC_ERR_ARG
    assert L.mc_comm_create(None, None, 1, 0, C.byref(C.c_void_p())) == mc.MC_ERR_ARG
    assert L.mc_march_rank(None, C.byref(p), None, None, None) == mc.MC_ERR_ARG


@pytest.mark.gpu
def test_reference_style_caller_with_a_device_list(mc):
    """Marching::set_devices({0, 0, ...}) -- 2, 3, 4 and 8 slabs on the one GPU -- hands out the single sweep's Poly_Data and
    soup bit for bit (five surfaces, one with a constraint and anisotropic scale, one lying in the lattice planes); with seed
    mode on, the list's first device sweeps the whole grid and the mesh is the single-device one.  All in C++ through the
    facade and the C ABI."""
    r = subprocess.run([str(build_sharded_caller(mc))], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SHARDED_OK" in r.stdout and "DIFFERENT" not in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 4, 8])
def test_march_sharded_equals_the_whole_sweep(mc, ctx, orc, n):
    """mc_march_sharded over a device list [0] * n: codes, soup (with normals) and -- MC_FLAG_INDEXED -- one Poly_Data equal
    the single sweep's bytes, which equal the oracle's; the slabs' offsets are the prefix sums of their counts; caller-chosen
    bounds (uneven, one EMPTY slab) give the same mesh."""
    import numpy as np
    eq, step = "x^2+y^2+z^2-1", float(np.float32(2.0) / np.float32(64))
    whole = ctx.march(eq, step)
    wv, wc = whole.vertices(), whole.codes()
    o = orc.march(eq, step, pow_mode=orc.POW_EXACT, want=7)
    assert np.array_equal(wc, o.codes) and np.array_equal(wv[:, :, :3].view(np.uint32), o.soup.view(np.uint32))
    sh = mc.Sharded([0] * n)
    try:
        n1 = mc.cells_per_axis(step)
        for bounds in (None, [0] + [n1 // 3] * (n - 1) + [n1]):    # near-equal slabs; uneven ones with n - 2 EMPTY slabs in the middle
            r = sh.march(eq, step, bounds=bounds)
            assert r.n_tris == whole.n_tris == o.n_tris and r.n_cells == whole.n_cells
            assert np.array_equal(r.codes(), wc)
            assert np.array_equal(r.vertices().view(np.uint32), wv.view(np.uint32))
            acc = 0
            for s, slab in zip(r.shards, r.slabs):
                assert s.tri_offset == acc and s.n_tris_total == whole.n_tris and (s.z_begin, s.z_end) == (slab.z_begin, slab.z_end)
                acc += slab.n_tris
            ri = sh.march(eq, step, flags=mc.FLAG_INDEXED | mc.FLAG_NO_EMIT, bounds=bounds)
            wi = ctx.march(eq, step, flags=mc.FLAG_INDEXED | mc.FLAG_NO_EMIT)
            v, t, nrm = ri.indexed()
            wvl, wtl, wn = wi.indexed()
            assert (ri.n_verts, ri.n_tris) == (wi.n_verts, wi.n_tris)
            assert np.array_equal(v.view(np.uint32), wvl.view(np.uint32)) and np.array_equal(t, wtl)
            assert np.array_equal(nrm.view(np.uint32), wn.view(np.uint32))
            assert ri.shards[-1].n_verts_total == wi.n_verts and ri.shards[0].vert_offset == 0
    finally:
        sh.close()


@pytest.mark.gpu
def test_march_sharded_headline_grid_counts(mc):
    """The 1025^3 sphere over a device list of 8 on the one GPU: the reference-scaled counts (x4 per
    doubling from 617 180 at 256) and slab offsets that tile the triangle list."""
    import numpy as np
    sh = mc.Sharded([0] * 8)
    try:
        r = sh.march("x^2+y^2+z^2-1", float(np.float32(2.0) / np.float32(1024)), flags=mc.FLAG_NORMALS)
        assert r.n_cells == 1025 ** 3 and r.n_tris == 9881660
        assert [s.tri_offset for s in r.shards] == list(np.cumsum([0] + [x.n_tris for x in r.slabs[:-1]]))
        assert all(x.n_tris > 1_000_000 for x in r.slabs)        # (a sphere's zones of equal height have equal area)
    finally:
        sh.close()


@pytest.mark.gpu
def test_march_sharded_reports_the_failing_slab(mc):
    import numpy as np
    sh = mc.Sharded([0, 0])
    try:
        with pytest.raises(mc.McError) as e:
            sh.march("x+", 0.25)
        assert e.value.code == mc.MC_ERR_EVAL and "slab" in str(e.value)
        with pytest.raises(mc.McError) as e:
            sh.march("x+y", 0.25, bounds=[0, 7, 3])
        assert e.value.code == mc.MC_ERR_ARG
    finally:
        sh.close()


RCCL_ONE_RANK = r'''
import sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import mc_amd as mc
ctx = mc.Context(0)
eq, step = "x^2+y^2+z^2-1", float(np.float32(2.0) / np.float32(48))
comm = mc.Comm(ctx, mc.Comm.new_id(), 1, 0)
whole = ctx.march(eq, step, flags=mc.FLAG_INDEXED | mc.FLAG_NORMALS)
wv, wt, _ = whole.indexed()
r, s = comm.march(eq, step, flags=mc.FLAG_INDEXED | mc.FLAG_NORMALS)
assert (r.n_tris, r.n_verts) == (whole.n_tris, whole.n_verts) and (s.tri_offset, s.vert_offset) == (0, 0)
assert (s.n_tris_total, s.n_verts_total) == (whole.n_tris, whole.n_verts)
v, t, _ = r.indexed()
assert np.array_equal(v.view(np.uint32), wv.view(np.uint32)) and np.array_equal(t, wt)
ctx.graph_build(eq, step, flags=mc.FLAG_NORMALS | mc.FLAG_NO_TIMING)
g = ctx.graph_replay(0.0)
for _ in range(70):                    # (more gathers than the communicator has slots: it drains and goes on)
    ctx.graph_replay_async(0.0)
    comm.gather_async(g.d_totals)
counts = comm.wait()
ctx.graph_wait()
assert counts.shape == (1, 2) and int(counts[0, 0]) == whole.n_tris and int(counts[0, 1]) == whole.n_active
comm.close()
ctx.close()
print("RCCL_ONE_RANK_OK", whole.n_tris)
'''


@pytest.mark.gpu
def test_rccl_communicator_with_one_rank(tmp_path):
    """mc_comm_* on the hardware there is: a world of ONE rank goes through librccl (dlopen, ncclGetUniqueId,
    ncclCommInitRank, ncclAllGather on the side stream) -- mc_march_rank's result is the whole sweep with offsets 0, and the
    asynchronous gather of a replayed graph's device-side counts returns that sweep's counts.  (More ranks than GPUs cannot
    share a device under RCCL; the N-rank logic is the gloo tests' above.)  In a process of its own, like an application
    that starts, makes its communicator and sweeps."""
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(RCCL_ONE_RANK)
    r = subprocess.run([sys.executable, str(script), str(ROOT)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
