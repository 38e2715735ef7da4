#!/usr/bin/env python3
"""bench.py -- marching-cubes voxel sweep throughput on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--grid-res R] [--equation EQ] [--no-cpu-baseline]

One "step" = one full pass of the hot path (classify -> scan -> emit, normals on) over the
headline workload of BASELINE.json: the sphere SDF x^2+y^2+z^2-1 on a 1024^3 grid
(grid_res 1024 -> 1025^3 cells swept, like the reference: marching.cpp:372-383), f analytic,
nothing to upload, all buffers resident in HBM.  With N > 1 (launched by torch.distributed.run,
one process per GPU) the SAME grid is sharded along Z into N contiguous slabs (strong scaling);
the only exchange is an RCCL all-gather of the per-rank triangle counts, from which every rank
derives its offset in the global triangle list (analytic f needs no halo: DESIGN.md).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (classify), measured live
with HIP events on the library's stream; `pipeline` gives the same figure for the whole
classify+scan+emit chain against SURVEY 8d's 2*C + 72*T bytes.  `cpu_baseline` times the CPU
oracle (oracle/, a port of the reference, all host cores) on a bounded Z-slab of the same
workload.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
EQ3 = "(x^2+y^2+z^2+(1/3)^2-(1/5)^2)^2-4*((1/2)*x-(2.36/6)*(1/5))^2-4*(1/3)^2*y^2"   # example_files/equation_3.txt (BASELINE config 3)
GYROID = "sin(x)*cos(y)+sin(y)*cos(z)+sin(z)*cos(x)"                                   # BASELINE config 4 (grammar extension E1)
PROFILE_TAG = "r04"


def traffic_profile(workload_key):
    """profiles/<tag>_pmc_traffic_<workload>.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
    (tools/profile_round.py), committed; None when this workload has not been profiled."""
    f = ROOT / "profiles" / f"{PROFILE_TAG}_pmc_traffic_{workload_key}.json"
    return f if f.exists() else None


def pmc_traffic(kernel, workload_key):
    """HBM bytes per launch of `kernel` from the committed PMC passes (FETCH_SIZE doubled per the gfx950 note in
    MI355X_MICROARCH.md); None if this workload has no profile."""
    f = traffic_profile(workload_key) if workload_key else None
    if f is None:
        return None
    try:
        return int(json.loads(f.read_text())[kernel]["hbm_bytes_per_launch_fetch_x2"])
    except Exception:
        return None


def spawn_ranks(n):
    """`python bench.py --gpus N` started plainly: become the launcher -- N fresh rank processes through
    torch.distributed.run, started BEFORE this process has touched a GPU (importing torch does not); rank 0's JSON line is
    relayed and the children's exit code returned."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env)
    return r.returncode


def cpu_baseline(eq, step, n1, budget_s=15.0):
    """Oracle (port of the reference) on the host cores, on a bounded slab of the same grid."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import pyoracle as orc
    # a one-GPU box hands this job a 16-CPU share whatever the affinity mask says
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1),
                int(os.environ.get("BENCH_CPU_THREADS", "16")))
    mid = n1 // 2
    t0 = time.perf_counter()
    probe_layers = max(1, min(cores, n1 - mid))
    orc.march(eq, step, 0.0, pow_mode=orc.POW_LIBM, want=orc.WANT_SOUP, z_begin=mid, z_end=mid + probe_layers,
              nthreads=cores)
    dt = max(time.perf_counter() - t0, 1e-3)
    layers = int(max(probe_layers, min(n1 - mid, budget_s / dt * probe_layers)))
    t0 = time.perf_counter()
    m = orc.march(eq, step, 0.0, pow_mode=orc.POW_LIBM, want=orc.WANT_SOUP, z_begin=mid, z_end=mid + layers,
                  nthreads=cores)
    dt = time.perf_counter() - t0
    ratio = None
    try:   # committed measurement: the port against the reference proper, one thread, on the machine of BASELINE.md's table
        rj = json.loads((ROOT / "profiles" / "r03_cpu_port_vs_reference.json").read_text())
        x = float(rj["sphere_port_over_reference_speed"])
        ratio = {"port_over_reference_speed_single_thread": x,
                 "reference_equivalent_mvoxels_per_s": round(m.n_cells / dt / 1e6 / x, 4),
                 "source": "profiles/r03_cpu_port_vs_reference.json (tools/cpu_ratio.py: oracle port vs BASELINE.md section 2, sphere)"}
    except Exception:
        pass
    return {"value": round(m.n_cells / dt / 1e6, 4), "unit": "Mvoxels/s", "cores": cores, "kind": "port", "ratio_to_reference": ratio,
            "sample": f"oracle/mc_oracle.c (libm powf, {cores} threads) on cell layers z=[{mid},{mid + layers}) of the "
                      f"same {n1}^3-cell grid: {m.n_cells} cells, {m.n_tris} triangles in {dt:.2f} s",
            "mtris_per_s": round(m.n_tris / dt / 1e6, 5)}


def cold_start(mc_amd, device):
    """Time to first mesh for an equation nobody has compiled, next to BASELINE.md section 2's 0.095 s (the reference's
    recalculate() of the sphere at grid_res 32, where Evaluator::set_equation costs nothing: evaluator.cpp:15-17).  A fresh
    context and an EMPTY code-object cache; wall-clock milliseconds of mc_march at grid_res 32, outside the timed region."""
    import tempfile
    old = os.environ.get("MC_JIT_CACHE")
    old_cold = os.environ.pop("MC_COLD_START", None)   # (MC_COLD_START=jit would make every first sweep wait for hiprtc: not what is measured here)
    out = {"grid_res": 32, "reference_s": 0.095, "reference_source": "BASELINE.md section 2 (sphere, grid_res 32, one CPU thread)"}
    with tempfile.TemporaryDirectory() as d:
        os.environ["MC_JIT_CACHE"] = d
        c = mc_amd.Context(device)
        try:
            tag = int(time.time() * 1e3) % 100000
            step = float(np.float32(2.0) / np.float32(32))
            fl = mc_amd.FLAG_NORMALS
            t0 = time.perf_counter()
            r = c.march(f"x^2+y^2+z^2-0.9{tag:05d}", step, flags=fl)          # loads the interpreter build's module too
            out["first_mesh_ms_fresh_context"] = round((time.perf_counter() - t0) * 1e3, 2)
            out["interpreted"] = bool(r.interpreted)
            t0 = time.perf_counter()
            r = c.march(f"x^2+y^2+z^2-0.8{tag:05d}", step, flags=fl)          # the next unseen equation on this context
            out["first_mesh_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
            t0 = time.perf_counter()
            ri = c.march(f"x^2+y^2+z^2-0.7{tag:05d}", step, flags=mc_amd.FLAG_INDEXED | mc_amd.FLAG_NO_EMIT)
            ri.indexed()
            out["first_indexed_mesh_ms"] = round((time.perf_counter() - t0) * 1e3, 2)   # what the facade's recalculate() + get_poly_data() costs
            eq = f"x^2+y^2+z^2-0.8{tag:05d}"
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 60.0 and c.march(eq, step, flags=fl).interpreted:
                time.sleep(0.02)
            out["specialised_kernels_take_over_after_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
            t0 = time.perf_counter()
            c.march(f"x^2+y^2+z^2-0.6{tag:05d}", step, flags=fl | mc_amd.FLAG_NO_INTERP)  # what every first sweep cost before: hiprtc
            out["first_mesh_ms_waiting_for_hiprtc"] = round((time.perf_counter() - t0) * 1e3, 1)
        finally:
            c.close()
            if old_cold is not None:
                os.environ["MC_COLD_START"] = old_cold
            if old is None:
                os.environ.pop("MC_JIT_CACHE", None)
            else:
                os.environ["MC_JIT_CACHE"] = old
    return out


def halo_check(ctx, mc_amd, torch, dist, eq, step, n1, zb, ze, rank, world, scale=1.0):
    """The one-slab halo of a sampled-field design, exchanged with RCCL send/recv and compared with
    what this design does instead (every rank evaluates its own top sample plane): identical bits.
    Outside the timed region; returns (plane bytes, exchange ms, identical)."""
    if world == 1:
        return None
    ax = (-1 + np.arange(n1 + 1, dtype=np.float64) * 0).astype(np.float32)  # rebuilt below with float adds
    v = np.float32(-1.0)
    for i in range(n1 + 1):
        ax[i] = v
        v = np.float32(v + np.float32(step))
    def plane(iz):
        xx, yy = np.meshgrid(ax, ax, indexing="xy")
        pts = np.stack([xx.ravel(), yy.ravel(), np.full(xx.size, ax[iz], np.float32)], axis=1)
        return ctx.eval_points(eq, (np.float32(scale) * pts).astype(np.float32))   # marching.cpp:211 scale * coordinate
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    mine_top = torch.from_numpy(plane(ze)).to(dev)        # sample plane above my last layer = next rank's first plane
    theirs = torch.empty_like(mine_top)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ops = []
    if rank + 1 < world:
        ops.append(dist.P2POp(dist.irecv, theirs, rank + 1))   # the halo a sampled-field design would need
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, torch.from_numpy(plane(zb)).to(dev), rank - 1))
    for w in dist.batch_isend_irecv(ops) if ops else []:
        w.wait()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    same = True if rank + 1 >= world else bool(torch.equal(theirs.view(torch.int32), mine_top.view(torch.int32)))
    flag = torch.tensor([1 if same else 0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return {"plane_bytes": int(mine_top.numel() * 4), "exchange_ms_incl_plane_eval": round(ms, 3),
            "recomputed_plane_identical_to_exchanged": bool(flag.item())}


def isosweep(args, torch, mc_amd, world, rank, local_rank, dist):
    """BASELINE config 5: Goursat form, grid_res 512, iso swept -0.7 -> -0.1, graph replay."""
    if world != 1:
        raise SystemExit("--mode isosweep is a single-GPU measurement")
    eq = "(x^2)^2+(y^2)^2+(z^2)^2-(x^2+y^2+z^2)"
    n = 512 if args.grid_res == 1024 else args.grid_res
    step = float(np.float32(2.0) / np.float32(n))
    depth = max(1, args.in_flight)
    ctxs = [mc_amd.Context(local_rank) for _ in range(depth)]
    ctx = ctxs[0]
    frames = max(args.steps, 2)
    isos = np.linspace(-0.7, -0.1, frames).astype(np.float32)
    t63 = mc_amd.FLAG_TILE63 if os.environ.get("MC_FORCE63") == "63" else 0   # developer A/B
    for c in ctxs:                                 # -0.4 has the most triangles: sizes the vertex buffers
        c.graph_build(eq, step, iso=-0.4, flags=mc_amd.FLAG_NORMALS | mc_amd.FLAG_NO_TIMING | mc_amd.FLAG_NO_INTERP | t63)

    def play(frame_isos, depth=depth):
        """Frame k goes to context k % depth (its own buffers and stream), so `depth` frames are in flight; EVERY frame's
        triangle count is read back -- when its context comes up again, depth - 1 frames later."""
        total, busy = 0, [False] * depth
        for k, iso in enumerate(frame_isos):
            c = ctxs[k % depth]
            if busy[k % depth]:
                total += c.graph_wait().n_tris
            c.graph_replay_async(float(iso))
            busy[k % depth] = True
        for j in range(depth):
            if busy[j]:
                total += ctxs[j].graph_wait().n_tris
        return total

    play(isos[: args.warmup])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tris = play(isos)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the same frames with ONE frame in flight (what an application that needs frame k before it starts k + 1 sees)
    t1 = time.perf_counter()
    tris1 = play(isos, 1)
    torch.cuda.synchronize()
    dt1 = time.perf_counter() - t1
    assert tris1 == tris
    # per-kernel times: the same frames once more, one at a time and kernel by kernel (mc_march), each kernel timed by the
    # start / stop timestamps of its own dispatch -- the figures rocprofv3's kernel trace reports, not event nodes that also
    # see the gaps between a graph's nodes (for kernels under 100 us those gaps were 10-55 % of the figure)
    kt = np.zeros(4)
    for iso in isos:
        r = ctx.march(eq, step, float(iso), flags=mc_amd.FLAG_NORMALS | mc_amd.FLAG_NO_INTERP | t63)
        kt += (r.ms_classify, r.ms_scan, r.ms_emit, r.ms_total)
    kt /= frames
    n1 = mc_amd.cells_per_axis(step)
    cells = float(n1) ** 3
    tpf = tris / frames
    print(json.dumps({"metric": "Mtris/s", "value": round(tris / dt / 1e6, 2), "unit": "Mtris/s", "n_gpus": 1, "steps": frames,
                      "warmup": args.warmup, "ms_per_step": round(dt / frames * 1e3, 4), "higher_is_better": True,
                      "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": f"iso sweep -0.7..-0.1 on {eq}, grid_res {n} ({n1}^3 cells), one hipGraph replay per frame, "
                                             f"{depth} frame(s) in flight (one context = one set of buffers each); every frame's "
                                             "triangle count is read back inside the timed region",
                                 "frames": frames, "triangles_total": int(tris), "in_flight": depth},
                      "mvoxels_per_s": round(n1 ** 3 * frames / dt / 1e6, 1),
                      "ms_per_step_one_in_flight": round(dt1 / frames * 1e3, 4), "mtris_per_s_one_in_flight": round(tris / dt1 / 1e6, 2),
                      "kernel_ms": {"classify": round(kt[0], 4), "scan": round(kt[1], 4), "emit": round(kt[2], 4), "gpu_total": round(kt[3], 4),
                                    "emit_kernel": "mc_emit" if r.emit_shared else "mc_emit_direct",
                                    "source": "start / stop timestamps of each kernel's own dispatch (hipExtModuleLaunchKernel events), the same "
                                              "frames swept one at a time behind the timed region"},
                      "roofline": {"bound": "hbm", "kernel": "mc_classify", "achieved": round(cells / (kt[0] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": round(cells / (kt[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   "traffic": pmc_traffic("mc_classify", "goursat512" if n == 512 else None),
                                   "algorithmic_bytes_per_launch": int(cells), "avg_launch_ms": round(kt[0], 4)},
                      "second_roofline": {"bound": "hbm", "kernel": "mc_emit" if r.emit_shared else "mc_emit_direct",
                                          "achieved": round(72 * tpf / (kt[2] * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": round(72 * tpf / (kt[2] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                          "traffic": pmc_traffic("mc_emit" if r.emit_shared else "mc_emit_direct", "goursat512" if n == 512 else None),
                                          "algorithmic_bytes_per_launch": int(72 * tpf), "avg_launch_ms": round(kt[2], 4)},
                      "pipeline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "formula": "2*C + 72*T (SURVEY 8d), mean frame",
                                   "algorithmic_bytes_per_launch": int(2 * cells + 72 * tpf),
                                   "achieved": round((2 * cells + 72 * tpf) / (kt[3] * 1e-3) / 1e9, 1),
                                   "frac": round((2 * cells + 72 * tpf) / (kt[3] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}), flush=True)
    for c in ctxs:
        c.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)    # (a sweep is < 0.5 ms: 100 steps keep the pipeline's fill and drain
    ap.add_argument("--warmup", type=int, default=10)    # below 2 % of the timed region; the whole run is the CPU baseline's 12 s)
    ap.add_argument("--grid-res", type=int, default=1024)
    ap.add_argument("--equation", default="x^2+y^2+z^2-1")
    ap.add_argument("--workload", choices=["sphere", "gyroid", "torus"], default="sphere",
                    help="sphere: the headline configuration.  gyroid: BASELINE config 4, sin x cos y + sin y cos z + sin z cos x "
                         "at 4 periods per axis -- needs the sin/cos grammar extension (not a reference input, DESIGN.md E1).  "
                         "torus: BASELINE config 3, example_files/equation_3.txt at grid_res 512")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--halo-check", action="store_true",
                    help="N > 1, outside the timed region: exchange the boundary sample plane with the Z neighbour (send/recv) and "
                         "compare it with the plane this rank recomputes -- the halo a sampled-field design would need (opt-in: "
                         "a diagnostic must not be able to stall the scaling measurement)")
    ap.add_argument("--no-balance", action="store_true",
                    help="N > 1: keep the equal-height Z slabs instead of re-cutting them by measured cost before the timed steps")
    ap.add_argument("--slab-of", type=int, default=0,
                    help="developer probe (single process): sweep only the middle slab of an N-way Z split, i.e. the per-rank "
                         "work of an N-GPU run, to see the fixed per-step costs that bound strong scaling")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every sweep kernel by kernel (mc_march) instead of replaying the captured hipGraph (mc_graph_replay)")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="independent sweeps kept in flight: step k runs on context k %% D (its own buffers, stream and captured "
                         "graph), so the ramp, tail and scan bubble of one sweep are filled by its neighbours.  1: one context, "
                         "every sweep behind the previous one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-normals", action="store_true")
    ap.add_argument("--mode", choices=["sweep", "isosweep"], default="sweep",
                    help="sweep: the headline metric (default).  isosweep: BASELINE config 5 -- animated iso sweep on the "
                         "512^3 Goursat surface, one captured hipGraph replayed per frame with a new iso value")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    import mc_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    multi = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"   # BENCH_FORCE_DIST: world_size 1 through the distributed path (test)
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # BENCH_BACKEND=gloo + BENCH_SINGLE_DEVICE=1: rehearsal of the N>1 path on a one-GPU box
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if os.environ.get("BENCH_SINGLE_DEVICE") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)

    if args.mode == "isosweep":
        return isosweep(args, torch, mc_amd, world, rank, local_rank, dist)
    eq = args.equation
    scale = (args.scale,) * 3
    if args.workload == "gyroid":
        mc_amd.set_extensions(mc_amd.EXT_TRIG)
        eq, scale = GYROID, (12.566371,) * 3
    if args.workload == "torus":
        eq = EQ3
        if args.grid_res == 1024:
            args.grid_res = 512
    step = float(np.float32(2.0) / np.float32(args.grid_res))
    n1 = mc_amd.cells_per_axis(step)
    zb, ze = mc_amd.shard_layers(n1, world, rank)
    if args.slab_of > 1 and world == 1:
        zb, ze = mc_amd.shard_layers(n1, args.slab_of, args.slab_of // 2)
    flags = 0 if args.no_normals else mc_amd.FLAG_NORMALS
    flags |= mc_amd.FLAG_NO_INTERP   # steady state: every sweep here runs the kernels specialised for the equation (the cold start has its own section)
    if os.environ.get("MC_FORCE63") == "63":     # developer A/B: 63-row classify tiles whatever the grid size
        flags |= mc_amd.FLAG_TILE63
    ctx = mc_amd.Context(local_rank)
    cdev = "cuda" if (multi and dist.get_backend() == "nccl") else "cpu"
    counts_dev = torch.zeros(world, dtype=torch.int64, device=cdev) if multi else None
    pending = []

    # Count-balanced Z repartition (untimed): equal-height slabs do not cost the same -- an equatorial slab of the
    # sphere has ~1.3x the undecided rows of the average one -- so each rank times its slab, the times are gathered,
    # and every rank re-cuts the layers so that the cumulative measured cost splits evenly (3 rounds).
    bounds = [mc_amd.shard_layers(n1, world, r)[0] for r in range(world)] + [n1]
    if world > 1 and not args.no_balance:
        tdev = torch.zeros(world, dtype=torch.float64, device=cdev)
        for _ in range(3):
            ctx.march(eq, step, 0.0, scale, flags=flags, z_begin=zb, z_end=ze)
            rr = ctx.march(eq, step, 0.0, scale, flags=flags, z_begin=zb, z_end=ze)
            dist.all_gather_into_tensor(tdev, torch.tensor([rr.ms_total], dtype=torch.float64, device=cdev))
            bounds = mc_amd.rebalance_layers(bounds, tdev.cpu().tolist())
            zb, ze = bounds[rank], bounds[rank + 1]

    # Steady state: the sweep (classify, scan, emit -- three kernel nodes; the totals reach pinned host memory through the
    # scan kernel's own store) is captured once as a hipGraph and replayed per step WITHOUT a host round trip:
    # K replays are enqueued back to back and the counts are read once at the end.  Two captures: a plain one for the
    # timed region, one with the per-kernel hipEvent nodes for the kernel times the roofline uses.
    # The steps are independent sweeps; step k runs on context k % depth -- each context has its own buffers, stream and
    # captured graph -- so `depth` sweeps are in flight and fill each other's ramps, tails and scan bubbles (one context:
    # 0.494 ms per 1025^3 sweep; three: 0.42; on a 1/8 slab 0.088 -> 0.061: tools/concurrent_probe2.py).
    depth = 1 if args.no_graph else max(1, args.in_flight)
    ctxs = [ctx] + [mc_amd.Context(local_rank) for _ in range(depth - 1)]
    # several sweeps in flight: tell the library (FLAG_BATCH: it then chooses for throughput where that differs from the
    # fastest single sweep -- mc_emit on small grids); the one-in-flight figure below is taken WITHOUT the flag, on a
    # context of its own, as a caller that issues one march() at a time would run
    lat_flags = flags
    if depth > 1:
        flags |= mc_amd.FLAG_BATCH
    r0 = None
    for c in ctxs:
        r0c = c.march(eq, step, 0.0, scale, flags=flags, z_begin=zb, z_end=ze)   # sizes every buffer; this rank's counts
        r0 = r0 or r0c
        if not args.no_graph:
            c.graph_build(eq, step, 0.0, scale, flags | mc_amd.FLAG_NO_TIMING, zb, ze)

    # N > 1, the path's one real exchange -- per-rank triangle counts -> global offsets -- without the host in the loop:
    # the sweep leaves its counts in device memory (mc_result.d_totals); a side stream, ordered behind the sweep, copies
    # them into this step's slot and feeds the RCCL all-gather; the next sweep is ordered behind that copy (its scan
    # clears the counts).  Only the fence at the end of the timed region waits for anything.
    lib_streams = sides = totals_devs = slots = None
    exchange_note = None
    # The exchange itself is the LIBRARY's (mc_comm_*: ncclAllGather through the RCCL the process has loaded, on a side stream
    # behind the sweep -- what a C++ caller with one process per GPU uses, include/mc_hip.h): one communicator per in-flight
    # context, their ids made by rank 0 and broadcast over the process group torch.distributed.run set up.  It is rehearsed
    # once before the timed steps and every rank must agree that it works; otherwise (or with BENCH_EXCHANGE=torch) the
    # same exchange goes through torch.distributed's all_gather_into_tensor as in rounds 1-3.
    comms = d_totals = None
    if multi and cdev == "cuda" and not args.no_graph and os.environ.get("BENCH_EXCHANGE", "library") == "library":
        ok = 1
        try:
            nb = mc_amd.COMM_ID_BYTES
            ids = torch.zeros(depth * nb + 1, dtype=torch.uint8, device=cdev)
            if rank == 0:
                try:
                    raw = b"".join(mc_amd.Comm.new_id() for _ in ctxs) + b"\x01"
                    ids.copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))
                except Exception as e:  # noqa: BLE001 -- e.g. librccl.so cannot be loaded: every rank learns it from the flag byte
                    exchange_note = f"torch (mc_comm_get_id failed: {e})"[:200]
            dist.broadcast(ids, 0)
            raw = ids.cpu().numpy().tobytes()
            if raw[-1] != 1:
                raise RuntimeError("rank 0 could not make a communicator id")
            comms = [mc_amd.Comm(c, raw[k * nb:(k + 1) * nb], world, rank) for k, c in enumerate(ctxs)]
            d_totals = [int(c.graph_replay(0.0).d_totals) for c in ctxs]
            for k, c in enumerate(ctxs):     # rehearsal: one replay + gather per context, checked against the host's count
                c.graph_replay_async(0.0)
                comms[k].gather_async(d_totals[k])
            for k, c in enumerate(ctxs):
                got = comms[k].wait()
                if int(got[rank, 0]) != int(c.graph_wait().n_tris):
                    raise RuntimeError(f"gathered count {int(got[rank, 0])} is not this rank's")
        except Exception as e:  # noqa: BLE001
            ok = 0
            exchange_note = exchange_note or f"torch (library exchange failed: {type(e).__name__}: {e})"[:200]
        agree = torch.tensor([ok], dtype=torch.int32, device=cdev)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if int(agree.item()) == 0:
            for cm in comms or []:
                try:
                    cm.close()
                except Exception:  # noqa: BLE001
                    pass
            comms = None
            exchange_note = exchange_note or "torch (another rank could not set the library exchange up)"
    if multi and cdev == "cuda" and not args.no_graph and comms is None:
        def raw_totals(c):   # zero-copy view of a context's {n_tris, n_active} words (same address sweep after sweep)
            class _Raw:
                __cuda_array_interface__ = {"shape": (2,), "typestr": "<i8", "data": (int(c.graph_replay(0.0).d_totals), False), "version": 2}
            return torch.as_tensor(_Raw(), device=f"cuda:{local_rank}")
        try:
            totals_devs = [raw_totals(c) for c in ctxs]
            lib_streams = [torch.cuda.ExternalStream(c.stream(), device=f"cuda:{local_rank}") for c in ctxs]
            sides = [torch.cuda.Stream(device=f"cuda:{local_rank}") for _ in ctxs]
            slots = torch.zeros(args.steps + args.warmup + 1, dtype=torch.int64, device=cdev)
        except Exception as e:  # noqa: BLE001 -- the exchange then goes through the host (below); the sweeps are the same
            exchange_note = f"host (device-side set-up failed: {type(e).__name__}: {e})"[:200]
            totals_devs = lib_streams = sides = slots = None
    step_no = [0]

    def one_step():
        i = step_no[0]
        step_no[0] += 1
        c = ctxs[i % depth]
        if args.no_graph:
            r = c.march(eq, step, 0.0, scale, flags=flags, z_begin=zb, z_end=ze)
        else:
            r = None
            c.graph_replay_async(0.0)
        if multi and comms is not None:
            comms[i % depth].gather_async(d_totals[i % depth])
        elif multi:
            if totals_devs is not None:
                lib_stream, side, totals_dev = lib_streams[i % depth], sides[i % depth], totals_devs[i % depth]
                side.wait_stream(lib_stream)
                with torch.cuda.stream(side):
                    slots[i:i + 1].copy_(totals_dev[0:1], non_blocking=True)
                    pending.append(dist.all_gather_into_tensor(counts_dev, slots[i:i + 1], async_op=True))
                lib_stream.wait_stream(side)
            else:   # gloo rehearsal / --no-graph: through the host
                n_tris = r.n_tris if r is not None else c.graph_wait().n_tris
                pending.append(dist.all_gather_into_tensor(counts_dev, torch.tensor([n_tris], dtype=torch.int64, device=cdev), async_op=True))
        return r

    def fence():
        r = None
        if not args.no_graph:
            for c in ctxs:
                rc = c.graph_wait()
                r = r or rc
        for wk in pending:
            wk.wait()
        pending.clear()
        if comms is not None and step_no[0] > 0:
            last = (step_no[0] - 1) % depth
            for k, cm in enumerate(comms):
                got = cm.wait()
                if k == last:
                    counts_dev.copy_(torch.from_numpy(got[:, 0].astype(np.int64)))
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        return r

    halo = None
    if world > 1 and args.halo_check:
        try:  # a diagnostic outside the timed region: never let it take the measurement down
            halo = halo_check(ctx, mc_amd, torch, dist, eq, step, n1, zb, ze, rank, world, scale[0])
        except Exception as e:  # noqa: BLE001
            halo = {"error": f"{type(e).__name__}: {e}"[:200]}
    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = one_step()
    rf = fence()
    elapsed = time.perf_counter() - t0
    r = rf if rf is not None else r

    # the same K steps with ONE sweep in flight (context 0 alone, each sweep behind the previous one), for comparison
    serial_ms = None
    kt1 = None
    ctx1 = None
    if not multi and not args.no_graph and depth > 1:
        ctx1 = mc_amd.Context(local_rank)   # (its own buffers and capture, without FLAG_BATCH)
        ctx1.march(eq, step, 0.0, scale, flags=lat_flags, z_begin=zb, z_end=ze)
        ctx1.graph_build(eq, step, 0.0, scale, lat_flags | mc_amd.FLAG_NO_TIMING, zb, ze)
        for _ in range(max(2, args.warmup)):
            ctx1.graph_replay_async(0.0)
        ctx1.graph_wait()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            ctx1.graph_replay_async(0.0)
        ctx1.graph_wait()
        torch.cuda.synchronize()
        serial_ms = (time.perf_counter() - t1) / args.steps * 1e3
        # ... and that caller's kernels, each alone: where the library launches a single sweep differently from a batch
        # (MC_FLAG_BATCH: on small grids the emit kernel of an expensive f), these are the single sweep's -- the ones a
        # rocprofv3 kernel trace of `bench.py --in-flight 1` shows (profiles/*_kernel_stats_<workload>.csv)
        kt1 = np.zeros(4)
        for _ in range(args.steps):
            r1 = ctx1.march(eq, step, 0.0, scale, flags=lat_flags, z_begin=zb, z_end=ze)
            kt1 += (r1.ms_classify, r1.ms_scan, r1.ms_emit, r1.ms_total)
        kt1 /= max(args.steps, 1)
        ctx1.close()

    # per-kernel GPU times for the roofline: the same K sweeps again, one at a time and kernel by kernel (mc_march), each
    # kernel timed by the start / stop timestamps of its own dispatch (hipExtModuleLaunchKernel events, on the library's
    # stream) -- the figures rocprofv3 --kernel-trace reports; right behind the timed region, same process, same buffers
    kt = np.zeros(4)
    for _ in range(args.steps):
        rk = ctx.march(eq, step, 0.0, scale, flags=flags, z_begin=zb, z_end=ze)
        kt += (rk.ms_classify, rk.ms_scan, rk.ms_emit, rk.ms_total)
    kt /= max(args.steps, 1)

    stats = torch.tensor([elapsed, float(r.n_cells), float(r.n_tris), *kt], dtype=torch.float64)
    if multi:
        stats = stats.to(cdev)
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        cells, tris = float(sm[1]), float(sm[2])
        kmax = mx[3:].cpu().numpy()
        offsets, total = mc_amd.exclusive_offsets(counts_dev.cpu().tolist())
        assert total == int(tris), (total, tris)
    else:
        cells, tris = float(r.n_cells), float(r.n_tris)
        kmax = kt

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        ms_cls, ms_scan, ms_emit, ms_tot = (float(x) for x in kmax)
        # algorithmic bytes (SURVEY 8d / DESIGN.md): classify writes 1 B per cell; emit reads 1 B per cell
        # and writes 72 B per triangle.  Per launch on this rank's slab (N=1: the whole grid).
        c_launch, t_launch = cells / world, tris / world
        cls_bytes = 1.0 * c_launch
        pipe_bytes = 2.0 * c_launch + 72.0 * t_launch
        cls_gbs = cls_bytes / (ms_cls * 1e-3) / 1e9 if ms_cls > 0 else 0.0
        pipe_gbs = pipe_bytes / (ms_tot * 1e-3) / 1e9 if ms_tot > 0 else 0.0
        emit_kernel = "mc_emit" if rk.emit_shared else "mc_emit_direct"
        # committed rocprofv3 PMC passes of this very workload (tools/profile_round.py), if there are any
        wkey = None
        if world == 1 and not args.no_normals and args.slab_of <= 1:
            wkey = {("x^2+y^2+z^2-1", 1024): "sphere1024", (EQ3, 512): "torus512", (GYROID, 1024): "gyroid1024"}.get((eq, args.grid_res))
        t_cls, t_emit = pmc_traffic("mc_classify", wkey), pmc_traffic(emit_kernel, wkey)
        out = {
            "metric": "Mvoxels/s", "value": round(cells / (elapsed / args.steps) / 1e6, 2), "unit": "Mvoxels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'sphere SDF' if eq == 'x^2+y^2+z^2-1' else args.workload} {eq}, grid_res {args.grid_res} "
                                   f"({n1}^3 cells), iso 0, scale {scale[0]:g}, normals {'off' if args.no_normals else 'on'}",
                       "cells": int(cells), "triangles": int(tris),
                       "parallelism": f"z-slab x{world}" if world > 1 else "single GPU",
                       "launch": ("kernel by kernel" if args.no_graph else
                                  f"hipGraph replays enqueued back to back, step k on context k % {depth} ({depth} independent sweeps "
                                  "in flight, each with its own buffers and stream), counts read once"),
                       "in_flight": depth,
                       "ms_per_step_one_in_flight": round(serial_ms, 4) if serial_ms is not None else None,
                       "count_exchange": (None if not multi else
                                          "library: mc_comm_gather_async (ncclAllGather over RCCL, side stream), device-side counts" if comms is not None
                                          else "rccl all_gather_into_tensor, device-side counts" + (f" [{exchange_note}]" if exchange_note else "") if totals_devs is not None
                                          else (exchange_note or "all_gather_into_tensor through the host")),
                       "z_bounds": bounds if world > 1 else None},
            "mtris_per_s": round(tris / (elapsed / args.steps) / 1e6, 3),
            "kernel_ms": {"classify": round(ms_cls, 4), "scan": round(ms_scan, 4), "emit": round(ms_emit, 4),
                          "gpu_total": round(ms_tot, 4), "emit_kernel": emit_kernel,
                          "source": f"start / stop timestamps of each kernel's own dispatch (hipExtModuleLaunchKernel events) in {args.steps} "
                                    "sweeps launched one at a time right behind the timed region"},
            # `roofline` is the DOMINANT kernel's (the one with the longer HIP-event time); the other one follows as
            # `second_roofline`.  classify: 1 B per cell written.  emit: what it must write (72 B per triangle); SURVEY 8d also
            # books 1 B/cell of code reads to it, which the record design never performs -- that byte only appears in the
            # pipeline figure below
            "roofline": None, "second_roofline": None,
            # the whole chain two ways: against SURVEY 8d's algorithmic bytes (which credit 1 B/cell of code READS this design
            # never performs), and against the bytes the counters actually see
            "pipeline": {"bound": "hbm", "achieved": round(pipe_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(pipe_gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(pipe_bytes),
                         "formula": "2*C + 72*T (SURVEY 8d)",
                         # achieved / frac above: the kernels of ONE sweep, alone (their HIP-event times).  The same bytes over
                         # the time a step takes in the timed region, where `in_flight` sweeps overlap:
                         "achieved_in_flight": round(pipe_bytes / (ms_step * 1e-3) / 1e9, 1),
                         "frac_in_flight": round(pipe_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "counter_bytes_per_launch": (t_cls + t_emit) if (t_cls is not None and t_emit is not None) else None,
                         "frac_by_counter_bytes": (round((t_cls + t_emit) / (ms_tot * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                                   if (t_cls is not None and t_emit is not None and ms_tot > 0) else None)},
        }
        r_cls = {"bound": "hbm", "kernel": "mc_classify", "achieved": round(cls_gbs, 1), "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": round(cls_gbs / HBM_PEAK_GBS, 4), "traffic": t_cls,
                 "traffic_source": (f"profiles/{PROFILE_TAG}_pmc_traffic_{wkey}.json (rocprofv3 --pmc, separate passes, per launch)"
                                    if t_cls is not None else None),
                 "algorithmic_bytes_per_launch": int(cls_bytes), "avg_launch_ms": round(ms_cls, 4)}
        emit_gbs = 72.0 * t_launch / (ms_emit * 1e-3) / 1e9 if ms_emit > 0 else 0.0
        r_emit = {"bound": "hbm", "kernel": emit_kernel, "achieved": round(emit_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": round(emit_gbs / HBM_PEAK_GBS, 4), "traffic": t_emit,
                  "traffic_source": (f"profiles/{PROFILE_TAG}_pmc_traffic_{wkey}.json (rocprofv3 --pmc, separate passes, per launch)"
                                     if t_emit is not None else None),
                  "algorithmic_bytes_per_launch": int(72.0 * t_launch), "avg_launch_ms": round(ms_emit, 4)}
        out["roofline"], out["second_roofline"] = (r_emit, r_cls) if ms_emit > ms_cls else (r_cls, r_emit)
        if serial_ms is not None:   # ONE sweep in flight: what a single march() / recalculate() call costs (steady state)
            out["kernel_ms_one_in_flight"] = {"classify": round(float(kt1[0]), 4), "scan": round(float(kt1[1]), 4), "emit": round(float(kt1[2]), 4),
                                              "gpu_total": round(float(kt1[3]), 4),
                                              "note": "the kernels of a caller that issues one sweep at a time (no MC_FLAG_BATCH); `kernel_ms` are the batch's"}
            out["ms_per_step_one_in_flight"] = round(serial_ms, 4)
            out["mvoxels_per_s_one_in_flight"] = round(cells / (serial_ms * 1e-3) / 1e6, 2)
            out["mtris_per_s_one_in_flight"] = round(tris / (serial_ms * 1e-3) / 1e6, 3)
        if halo is not None:
            out["halo"] = halo
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(eq, step, n1)
            try:   # outside the timed region, never in the way of the headline
                out["cold_start"] = cold_start(mc_amd, local_rank)
                out["cold_start_ms"] = out["cold_start"].get("first_mesh_ms")
            except Exception as e:  # noqa: BLE001
                out["cold_start"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        print(json.dumps(out), flush=True)
    for cm in comms or []:
        cm.close()
    for c in ctxs:
        c.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
