"""marching-cube-for-implicit-surfaces_amd -- MI355X-native marching-cubes hot path.

Thin ctypes binding of libmc_hip.so (C ABI: include/mc_hip.h).  The compute is
hand-written HIP for gfx950 inside that library; this module only mirrors the
reference's operator surface (Evaluator / Marching, Source/evaluator.h:24-86,
Source/marching.h:72-157) for tests and bench.py.  There is no CPU fallback:
if the library is missing or no GPU is visible, calls raise.

The directory name is not a Python identifier; load it with importlib (see
mc_amd.py at the repository root, which does exactly that).
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
# MC_AMD_DEV_LIB=1 (tools/ab.py only) loads the developer build, which honours the MC_JIT_EXTRA / MC_WPB_* hooks
LIB_PATH = _HERE / ("libmc_hip_dev.so" if os.environ.get("MC_AMD_DEV_LIB") == "1" else "libmc_hip.so")

MC_OK, MC_ERR_PARSE, MC_ERR_EVAL, MC_ERR_STEP, MC_ERR_ARG, MC_ERR_HIP, MC_ERR_NOMEM, MC_ERR_OVERFLOW = range(8)
FLAG_NORMALS, FLAG_KEEP_CODES, FLAG_NO_EMIT, FLAG_TILE1, FLAG_INDEXED, FLAG_NO_CULL, FLAG_NO_TIMING = 1, 2, 4, 8, 32, 64, 128
FLAG_EMIT_DIRECT, FLAG_EMIT_SHARED, FLAG_SEAM, FLAG_TILE63 = 256, 512, 1024, 2048
FLAG_BATCH = 4096   # several sweeps in flight (one context each): choose for throughput where that differs from the fastest single sweep
FLAG_ORDER_Z, FLAG_ORDER_MIDDLE_OUT = 8192, 16384   # diagnostic: force classify's launch order of the layers
FLAG_INTERP, FLAG_NO_INTERP = 32768, 65536           # cold start: sweep with the interpreter build / wait for hiprtc
COMM_ID_BYTES = 128

# every symbol include/mc_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "mc_abi_version", "mc_last_error", "mc_device_count", "mc_expr_check", "mc_expr_validate", "mc_expr_dump",
    "mc_expr_debug_eval_host", "mc_context_create", "mc_context_destroy", "mc_eval_points", "mc_march",
    "mc_march_simple", "mc_jit_precompile", "mc_copy_vertices", "mc_copy_soup", "mc_copy_soup_normals", "mc_copy_codes", "mc_copy_indexed", "mc_cells_per_axis", "mc_graph_build",
    "mc_graph_replay", "mc_graph_replay_async", "mc_graph_wait", "mc_stream", "mc_set_constraint", "mc_use_constraint", "mc_set_extensions",
    "mc_set_seed", "mc_seed_mode", "mc_context_set_extensions", "mc_index_rebase",
    "mc_shard_layers", "mc_march_sharded", "mc_copy_sharded_vertices", "mc_copy_sharded_indexed", "mc_copy_sharded_codes",
    "mc_expr_debug_interp_host", "mc_comm_get_id", "mc_comm_create", "mc_comm_destroy", "mc_march_rank", "mc_comm_gather_async", "mc_comm_wait",
]


class McParams(C.Structure):
    _fields_ = [("equation", C.c_char_p), ("step", C.c_float), ("iso", C.c_float), ("scale", C.c_float * 3),
                ("flags", C.c_uint32), ("z_begin", C.c_int32), ("z_end", C.c_int32)]


class McResult(C.Structure):
    _fields_ = [("cells_per_axis", C.c_int32), ("z_begin", C.c_int32), ("z_end", C.c_int32),
                ("n_cells", C.c_uint64), ("n_active", C.c_uint64), ("n_tris", C.c_uint64),
                ("d_vertices", C.c_void_p), ("d_codes", C.c_void_p), ("code_pitch", C.c_uint64),
                ("ms_classify", C.c_float), ("ms_scan", C.c_float), ("ms_emit", C.c_float), ("ms_total", C.c_float),
                ("code_main_cells", C.c_int32), ("d_codes_tail", C.c_void_p),
                ("n_verts", C.c_uint64), ("d_vertex_list", C.c_void_p), ("d_tri_list", C.c_void_p), ("d_vertex_normals", C.c_void_p),
                ("ms_index", C.c_float), ("d_totals", C.c_void_p), ("emit_shared", C.c_int32), ("interpreted", C.c_int32)]


class McShard(C.Structure):
    _fields_ = [("z_begin", C.c_int32), ("z_end", C.c_int32), ("tri_offset", C.c_uint64), ("vert_offset", C.c_uint64),
                ("n_tris_total", C.c_uint64), ("n_verts_total", C.c_uint64)]


class McError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmc_hip error {code}: {msg}")
        self.code = code


_lib = None


def build(force=False, verbose=False):
    """Compile libmc_hip.so in-tree (hipcc --offload-arch=gfx950)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_mc_build", _HERE / "build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(force=force, verbose=verbose)


def lib():
    """Load libmc_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} is missing: run `python __graft_entry__.py build` (no CPU fallback exists)")
        os.environ.setdefault("MC_JIT_CACHE", str(_HERE / "_jit_cache"))
        (_HERE / "_jit_cache").mkdir(exist_ok=True)
        L = C.CDLL(str(LIB_PATH))
        L.mc_last_error.restype = C.c_char_p
        L.mc_expr_check.argtypes = [C.c_char_p]
        L.mc_expr_validate.argtypes = [C.c_char_p]
        L.mc_expr_dump.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.mc_expr_dump.restype = C.c_size_t
        L.mc_expr_debug_eval_host.argtypes = [C.c_char_p, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.mc_expr_debug_interp_host.argtypes = [C.c_char_p, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.mc_jit_precompile.argtypes = [C.c_char_p, C.POINTER(C.c_size_t)]
        L.mc_context_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.mc_context_destroy.argtypes = [C.c_void_p]
        L.mc_context_destroy.restype = None
        L.mc_eval_points.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.mc_march.argtypes = [C.c_void_p, C.POINTER(McParams), C.POINTER(McResult)]
        L.mc_march_simple.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_float, C.c_uint32, C.POINTER(McResult)]
        L.mc_copy_vertices.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.mc_copy_soup.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.mc_copy_soup_normals.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.mc_copy_codes.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.mc_copy_indexed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
        L.mc_graph_replay_async.argtypes = [C.c_void_p, C.c_float]
        L.mc_graph_wait.argtypes = [C.c_void_p, C.POINTER(McResult)]
        L.mc_stream.argtypes = [C.c_void_p]
        L.mc_stream.restype = C.c_void_p
        L.mc_cells_per_axis.argtypes = [C.c_float]
        L.mc_set_constraint.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_float]
        L.mc_use_constraint.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.mc_graph_build.argtypes = [C.c_void_p, C.POINTER(McParams)]
        L.mc_graph_replay.argtypes = [C.c_void_p, C.c_float, C.POINTER(McResult)]
        L.mc_context_set_extensions.argtypes = [C.c_void_p, C.c_int]
        L.mc_index_rebase.argtypes = [C.c_void_p, C.c_uint64]
        L.mc_shard_layers.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.mc_shard_layers.restype = None
        L.mc_march_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(McParams), C.POINTER(C.c_int32), C.POINTER(McResult), C.POINTER(McShard)]
        L.mc_copy_sharded_vertices.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_uint64]
        L.mc_copy_sharded_indexed.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64]
        L.mc_copy_sharded_codes.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_uint64]
        L.mc_comm_get_id.argtypes = [C.c_void_p]
        L.mc_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.mc_comm_destroy.argtypes = [C.c_void_p]
        L.mc_comm_destroy.restype = None
        L.mc_march_rank.argtypes = [C.c_void_p, C.POINTER(McParams), C.POINTER(C.c_int32), C.POINTER(McResult), C.POINTER(McShard)]
        L.mc_comm_gather_async.argtypes = [C.c_void_p, C.c_void_p]
        L.mc_comm_wait.argtypes = [C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _check(rc):
    if rc != MC_OK:
        raise McError(rc, lib().mc_last_error().decode(errors="replace"))


# ---- expression layer (Evaluator) ------------------------------------------------------
EXT_TRIG = 1


def set_extensions(ext: int) -> int:
    """Process-wide grammar extensions (0 = the reference's grammar; EXT_TRIG adds sin(...) / cos(...))."""
    lib().mc_set_extensions.restype = C.c_uint
    return lib().mc_set_extensions(C.c_uint(ext))


def expr_check(eq: str) -> bool:
    """Evaluator::set_equation accept/reject (evaluator.cpp:15-17, :139-237)."""
    return bool(lib().mc_expr_check(eq.encode()))


def expr_validate(eq: str) -> int:
    """MC_OK / MC_ERR_PARSE / MC_ERR_EVAL without raising."""
    return lib().mc_expr_validate(eq.encode())


def expr_dump(eq: str) -> str:
    n = lib().mc_expr_dump(eq.encode(), None, 0)
    if n == 0:
        raise McError(MC_ERR_PARSE, lib().mc_last_error().decode())
    buf = C.create_string_buffer(n + 1)
    lib().mc_expr_dump(eq.encode(), buf, n + 1)
    return buf.value.decode()


def expr_debug_eval_host(eq: str, x, y, z) -> float:
    out = C.c_float()
    _check(lib().mc_expr_debug_eval_host(eq.encode(), x, y, z, C.byref(out)))
    return out.value


def expr_debug_interp_host(eq: str, x, y, z) -> float:
    """The same point through the interpreter build's program for `eq`, walked on the host (raises McError when the equation
    does not fit the interpreter's tables)."""
    out = C.c_float()
    _check(lib().mc_expr_debug_interp_host(eq.encode(), x, y, z, C.byref(out)))
    return out.value


def jit_precompile(eq: str) -> int:
    """hiprtc-compile the kernels for `eq` (no GPU needed); returns the code-object size."""
    n = C.c_size_t()
    _check(lib().mc_jit_precompile(eq.encode(), C.byref(n)))
    return n.value


def cells_per_axis(step) -> int:
    return lib().mc_cells_per_axis(C.c_float(step))


def device_count() -> int:
    return lib().mc_device_count()


class Result:
    """Host view of one sweep (counts + timings); arrays are fetched on demand."""

    def __init__(self, ctx, r: McResult):
        self._ctx = ctx
        for k, _ in McResult._fields_:
            setattr(self, k, getattr(r, k))

    def vertices(self) -> np.ndarray:
        """(n_tris, 3, 6) float32: x,y,z,nx,ny,nz."""
        a = np.empty((self.n_tris, 3, 6), dtype=np.float32)
        if self.n_tris:
            _check(lib().mc_copy_vertices(self._ctx._h, a.ctypes.data, self.n_tris))
        return a

    def soup(self) -> np.ndarray:
        """(n_tris, 3, 3) float32 positions, the reference's pre-dedup triangle soup."""
        a = np.empty((self.n_tris, 3, 3), dtype=np.float32)
        if self.n_tris:
            _check(lib().mc_copy_soup(self._ctx._h, a.ctypes.data, self.n_tris))
        return a

    def soup_normals(self) -> np.ndarray:
        """(n_tris, 3, 3) float32: the unit normals of soup()'s vertices."""
        a = np.empty((self.n_tris, 3, 3), dtype=np.float32)
        if self.n_tris:
            _check(lib().mc_copy_soup_normals(self._ctx._h, a.ctypes.data, self.n_tris))
        return a

    def indexed(self):
        """(vertex_list (V,3) f32, tri_list (T,3) u32, normals (V,3) f32): the reference's Poly_Data, welded on the GPU
        (FLAG_INDEXED), and the drawer's CalculateNormal."""
        v = np.empty((self.n_verts, 3), dtype=np.float32)
        n = np.empty((self.n_verts, 3), dtype=np.float32)
        t = np.empty((self.n_tris, 3), dtype=np.uint32)
        _check(lib().mc_copy_indexed(self._ctx._h, v.ctypes.data, t.ctypes.data, n.ctypes.data, self.n_verts, self.n_tris))
        return v, t, n

    def codes(self) -> np.ndarray:
        """(n_cells,) uint8 raw cube codes in sweep order (x fastest)."""
        a = np.empty(self.n_cells, dtype=np.uint8)
        if self.n_cells:
            _check(lib().mc_copy_codes(self._ctx._h, a.ctypes.data, self.n_cells))
        return a


class Context:
    """One GPU: stream, buffers and the cache of equations compiled to kernels."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        _check(lib().mc_context_create(device, C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            lib().mc_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _params(self, equation, step, iso, scale, flags, z_begin, z_end):
        p = McParams()
        p.equation = equation.encode()
        p.step = step
        p.iso = iso
        p.scale = (C.c_float * 3)(*scale)
        p.flags = flags
        p.z_begin = z_begin
        p.z_end = z_end
        return p

    def march(self, equation, step, iso=0.0, scale=(1.0, 1.0, 1.0), flags=FLAG_NORMALS | FLAG_KEEP_CODES, z_begin=0,
              z_end=-1) -> Result:
        """Marching::recalculate() full sweep (marching.cpp:368-384) on the GPU."""
        p = self._params(equation, step, iso, scale, flags, z_begin, z_end)
        r = McResult()
        _check(lib().mc_march(self._h, C.byref(p), C.byref(r)))
        return Result(self, r)

    def march_grid(self, equation, grid_res, iso=0.0, flags=FLAG_NORMALS | FLAG_KEEP_CODES) -> Result:
        """north-star form march(equation, grid_res, iso): step = 2.0f/grid_res."""
        r = McResult()
        _check(lib().mc_march_simple(self._h, equation.encode(), grid_res, iso, flags, C.byref(r)))
        return Result(self, r)

    def set_constraint(self, i: int, lhs: str, op: str, rhs: float, use: bool = True):
        """Marching::set_constraint(i, lhs, op, rhs) + use_constraint(i, use) (marching.cpp:173-207): cells with a
        corner where `lhs op rhs` is false are skipped by the following sweeps of this context."""
        _check(lib().mc_set_constraint(self._h, i, lhs.encode(), op.encode(), C.c_float(rhs)))
        _check(lib().mc_use_constraint(self._h, i, 1 if use else 0))

    def set_seed(self, x: float, y: float, z: float):
        """Marching::set_seed (marching.cpp:125-137)."""
        lib().mc_set_seed.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        _check(lib().mc_set_seed(self._h, x, y, z))

    def seed_mode(self, on: bool):
        """Marching::seed_mode (marching.cpp:115-118): keep only the surface reached from the seed's cell."""
        lib().mc_seed_mode.argtypes = [C.c_void_p, C.c_int]
        _check(lib().mc_seed_mode(self._h, 1 if on else 0))

    def use_constraint(self, i: int, use: bool):
        _check(lib().mc_use_constraint(self._h, i, 1 if use else 0))

    def index_rebase(self, vertex_offset: int):
        """FLAG_INDEXED | FLAG_SEAM: add the number of vertices owned by the slabs below to the last sweep's tri_list."""
        _check(lib().mc_index_rebase(self._h, vertex_offset))

    def set_extensions(self, ext: int):
        """Grammar extensions of THIS context (ext >= 0), whatever the process-wide setting; ext < 0: follow it again."""
        _check(lib().mc_context_set_extensions(self._h, ext))

    def eval_points(self, equation, pts) -> np.ndarray:
        """Evaluator::evaluate for many points (evaluator.cpp:53), computed on the GPU."""
        pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
        out = np.empty(len(pts), dtype=np.float32)
        _check(lib().mc_eval_points(self._h, equation.encode(), pts.ctypes.data, len(pts), out.ctypes.data))
        return out

    def graph_build(self, equation, step, iso=0.0, scale=(1.0, 1.0, 1.0), flags=FLAG_NORMALS, z_begin=0, z_end=-1):
        p = self._params(equation, step, iso, scale, flags, z_begin, z_end)
        _check(lib().mc_graph_build(self._h, C.byref(p)))

    def graph_replay(self, iso) -> Result:
        r = McResult()
        _check(lib().mc_graph_replay(self._h, iso, C.byref(r)))
        return Result(self, r)

    def graph_replay_async(self, iso):
        """Enqueue one replay without waiting for it (graph_wait collects the last one)."""
        _check(lib().mc_graph_replay_async(self._h, iso))

    def graph_wait(self) -> Result:
        r = McResult()
        _check(lib().mc_graph_wait(self._h, C.byref(r)))
        return Result(self, r)

    def stream(self) -> int:
        """The context's hipStream_t as an integer (torch.cuda.ExternalStream(ctx.stream()))."""
        return int(lib().mc_stream(self._h) or 0)


def _bounds_arg(bounds, n):
    if bounds is None:
        return None
    if len(bounds) != n + 1:
        raise ValueError(f"bounds must hold {n + 1} layer indices")
    return (C.c_int32 * (n + 1))(*[int(b) for b in bounds])


class ShardedResult:
    """Host view of one multi-device sweep (mc_march_sharded): the slabs' Results in slab order, their offsets, and the
    concatenated arrays -- what a single sweep's Result gives."""

    def __init__(self, sharded, results, shards):
        self._s = sharded
        self.slabs = results
        self.shards = shards
        self.n_tris = sum(r.n_tris for r in results)
        self.n_verts = sum(r.n_verts for r in results)
        self.n_cells = sum(r.n_cells for r in results)
        self.n_active = sum(r.n_active for r in results)

    def vertices(self) -> np.ndarray:
        a = np.empty((self.n_tris, 3, 6), dtype=np.float32)
        if self.n_tris:
            _check(lib().mc_copy_sharded_vertices(self._s._arr, len(self.slabs), a.ctypes.data, self.n_tris))
        return a

    def indexed(self):
        v = np.empty((self.n_verts, 3), dtype=np.float32)
        n = np.empty((self.n_verts, 3), dtype=np.float32)
        t = np.empty((self.n_tris, 3), dtype=np.uint32)
        _check(lib().mc_copy_sharded_indexed(self._s._arr, len(self.slabs), v.ctypes.data, t.ctypes.data, n.ctypes.data, self.n_verts, self.n_tris))
        return v, t, n

    def codes(self) -> np.ndarray:
        a = np.empty(self.n_cells, dtype=np.uint8)
        if self.n_cells:
            _check(lib().mc_copy_sharded_codes(self._s._arr, len(self.slabs), a.ctypes.data, self.n_cells))
        return a


class Sharded:
    """One process, a device list: one context per entry (a device may appear several times), the sweep's cell layers cut
    into one Z slab per context and swept at once (mc_march_sharded; SURVEY 8b "device list")."""

    def __init__(self, devices):
        self.ctxs = [Context(int(d)) for d in devices]
        self._arr = (C.c_void_p * len(self.ctxs))(*[c._h for c in self.ctxs])

    def close(self):
        for c in self.ctxs:
            c.close()
        self.ctxs = []

    def march(self, equation, step, iso=0.0, scale=(1.0, 1.0, 1.0), flags=FLAG_NORMALS | FLAG_KEEP_CODES, z_begin=0, z_end=-1,
              bounds=None) -> ShardedResult:
        n = len(self.ctxs)
        p = self.ctxs[0]._params(equation, step, iso, scale, flags, z_begin, z_end)
        res = (McResult * n)()
        sh = (McShard * n)()
        _check(lib().mc_march_sharded(self._arr, n, C.byref(p), _bounds_arg(bounds, n), res, sh))
        return ShardedResult(self, [Result(c, r) for c, r in zip(self.ctxs, res)], list(sh))


class Comm:
    """One process per device: an RCCL communicator over `world` processes for the sweep's one exchange, the per-rank
    counts (mc_comm_*).  Rank 0 calls Comm.new_id() and hands the 128 bytes to the others (any channel)."""

    @staticmethod
    def new_id() -> bytes:
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        _check(lib().mc_comm_get_id(buf))
        return bytes(buf)

    def __init__(self, ctx: "Context", comm_id: bytes, world: int, rank: int):
        assert len(comm_id) == COMM_ID_BYTES
        self.ctx, self.world, self.rank = ctx, world, rank
        h = C.c_void_p()
        idb = (C.c_uint8 * COMM_ID_BYTES)(*comm_id)
        _check(lib().mc_comm_create(ctx._h, idb, world, rank, C.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            lib().mc_comm_destroy(self._h)
            self._h = None

    def march(self, equation, step, iso=0.0, scale=(1.0, 1.0, 1.0), flags=FLAG_NORMALS | FLAG_KEEP_CODES, z_begin=0, z_end=-1, bounds=None):
        """This rank's slab + the all-gather of the counts (collective) -> (Result, McShard)."""
        p = self.ctx._params(equation, step, iso, scale, flags, z_begin, z_end)
        r, sh = McResult(), McShard()
        _check(lib().mc_march_rank(self._h, C.byref(p), _bounds_arg(bounds, self.world), C.byref(r), C.byref(sh)))
        return Result(self.ctx, r), sh

    def gather_async(self, d_totals: int):
        """Enqueue the all-gather of a sweep's device-side counts (Result.d_totals) behind the context's stream."""
        _check(lib().mc_comm_gather_async(self._h, C.c_void_p(d_totals)))

    def wait(self) -> np.ndarray:
        """Block until every gather enqueued is done; (world, 2) uint64 {n_tris, n_active} per rank of the last one."""
        out = np.zeros((self.world, 2), dtype=np.uint64)
        _check(lib().mc_comm_wait(self._h, out.ctypes.data))
        return out


# ---- Z-slab sharding across GPUs (one process per GPU; host logic only) -----------------
def shard_layers(n_layers: int, world: int, rank: int):
    """Contiguous near-equal range [z_begin, z_end) of cell layers for `rank`.

    Concatenating the ranks' triangle lists in rank order reproduces the single-GPU /
    reference emission order, because the sweep is z-major (marching.cpp:375)."""
    base, rem = divmod(n_layers, world)
    b = rank * base + min(rank, rem)
    return b, b + base + (1 if rank < rem else 0)


def rebalance_layers(bounds, costs):
    """Count-balanced Z repartition (host logic only).  `bounds` (world+1 ascending layer indices) are the slabs just
    swept, `costs[i]` what slab i cost (e.g. its GPU milliseconds).  Treating the cost as spread evenly over a slab's
    layers, returns new bounds that cut the cumulative cost into equal parts; every slab keeps at least one layer.
    Deterministic: every rank computes the same answer from the same gathered costs."""
    world = len(bounds) - 1
    n = bounds[-1] - bounds[0]
    if world <= 1 or n < world:
        return list(bounds)
    dens = [float(c) / max(1, bounds[i + 1] - bounds[i]) for i, c in enumerate(costs)]
    total = sum(d * (bounds[i + 1] - bounds[i]) for i, d in enumerate(dens))
    if not total > 0:
        return list(bounds)
    new = [bounds[0]]
    acc, slab = 0.0, 0
    z = bounds[0]
    for k in range(1, world):
        target = total * k / world
        while slab < world and acc + dens[slab] * (bounds[slab + 1] - z) < target:
            acc += dens[slab] * (bounds[slab + 1] - z)
            slab += 1
            z = bounds[slab] if slab < world else bounds[-1]
        if slab >= world:
            cut = bounds[-1]
        else:
            step_layers = (target - acc) / dens[slab] if dens[slab] > 0 else 0.0
            cut = int(round(z + step_layers))
        cut = max(cut, new[-1] + 1)                       # at least one layer per slab
        cut = min(cut, bounds[-1] - (world - k))          # ... for the slabs still to come as well
        new.append(cut)
    new.append(bounds[-1])
    return new


def exclusive_offsets(counts):
    """Per-rank triangle offsets from the all-gathered per-rank counts."""
    out, acc = [], 0
    for c in counts:
        out.append(acc)
        acc += int(c)
    return out, acc
