"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/mc_oracle.h).  The product never does.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
POW_LIBM, POW_EXACT = 0, 1
WANT_CODES, WANT_SOUP, WANT_NORMALS = 1, 2, 4


class _Token(C.Structure):
    _fields_ = [("type", C.c_int), ("ch", C.c_char), ("num", C.c_float)]


class _Expr(C.Structure):
    _fields_ = [("n", C.c_int), ("cap", C.c_int), ("tok", C.POINTER(_Token))]


class _Constraint(C.Structure):
    _fields_ = [("lhs", C.c_char_p), ("op", C.c_int), ("rhs", C.c_float)]


class _Mesh(C.Structure):
    _fields_ = [
        ("n1", C.c_int),
        ("n_cells", C.c_uint64),
        ("n_active", C.c_uint64),
        ("n_tris", C.c_uint64),
        ("n_amb", C.c_uint64),
        ("n_flipped", C.c_uint64),
        ("fnv_codes", C.c_uint64),
        ("fnv_soup", C.c_uint64),
        ("codes", C.POINTER(C.c_uint8)),
        ("soup", C.POINTER(C.c_float)),
        ("normals", C.POINTER(C.c_float)),
    ]


def build(force=False):
    """Compile liboracle.so (and _ref/check_tables where the reference exists)."""
    so = _HERE / "liboracle.so"
    srcs = [_HERE / "mc_oracle.c", _HERE / "mc_oracle_seed.cpp", _HERE / "mc_oracle_weld.cpp", _HERE / "mc_oracle.h", _HERE.parent / "include" / "mc_tables_data.h",
            _HERE.parent / "include" / "mc_trig.h"]
    if force or not so.exists() or so.stat().st_mtime < max(p.stat().st_mtime for p in srcs):
        subprocess.run(["make", "-C", str(_HERE)], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = build()
        L = C.CDLL(str(so))
        L.orc_tokenize.argtypes = [C.c_char_p, C.POINTER(_Expr)]
        L.orc_tokenize.restype = C.c_int
        L.orc_expr_free.argtypes = [C.POINTER(_Expr)]
        L.orc_evaluate.argtypes = [C.POINTER(_Expr), C.c_float, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_float)]
        L.orc_evaluate.restype = C.c_int
        L.orc_cells_per_axis.argtypes = [C.c_float]
        L.orc_cells_per_axis.restype = C.c_int
        L.orc_axis_coords.argtypes = [C.c_float, C.POINTER(C.c_float), C.c_int]
        L.orc_axis_coords.restype = C.c_int
        L.orc_march.argtypes = [C.c_char_p, C.c_float, C.c_float, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.POINTER(_Mesh)]
        L.orc_march.restype = C.c_int
        L.orc_march_constrained.argtypes = [C.c_char_p, C.c_float, C.c_float, C.POINTER(C.c_float), C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_int, C.POINTER(_Constraint), C.c_int, C.POINTER(_Mesh)]
        L.orc_march_constrained.restype = C.c_int
        L.orc_mesh_free.argtypes = [C.POINTER(_Mesh)]
        L.orc_fnv1a.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        L.orc_fnv1a.restype = C.c_uint64
        for name, ty in (("orc_tri_rows", C.c_uint64), ("orc_tri_counts", C.c_uint8), ("orc_amb_faces", C.c_uint8),
                         ("orc_face_corners", C.c_uint16), ("orc_edge_corners", C.c_uint8)):
            getattr(L, name).restype = C.POINTER(ty)
        _lib = L
    return _lib


def tokenize(eq: str) -> bool:
    e = _Expr()
    ok = lib().orc_tokenize(eq.encode(), C.byref(e))
    lib().orc_expr_free(C.byref(e))
    return bool(ok)


def tokens(eq: str):
    """Token list as (type, char, num) tuples, or None when rejected."""
    e = _Expr()
    if not lib().orc_tokenize(eq.encode(), C.byref(e)):
        return None
    out = [(e.tok[i].type, e.tok[i].ch.decode(), e.tok[i].num) for i in range(e.n)]
    lib().orc_expr_free(C.byref(e))
    return out


def march_seed(eq: str, step: float, seed, iso: float = 0.0, scale=(1.0, 1.0, 1.0), pow_mode=POW_LIBM,
               want=WANT_SOUP) -> "Mesh":
    """Seed mode (marching.cpp:310-331): the mesh reached from the seed's cell, breadth-first order."""
    L = lib()
    L.orc_march_seed.argtypes = [C.c_char_p, C.c_float, C.c_float, C.POINTER(C.c_float), C.c_int, C.c_int,
                                 C.POINTER(C.c_float), C.POINTER(_Mesh)]
    L.orc_march_seed.restype = C.c_int
    m = _Mesh()
    r = L.orc_march_seed(eq.encode(), C.c_float(step), C.c_float(iso), (C.c_float * 3)(*scale), pow_mode, want,
                         (C.c_float * 3)(*seed), C.byref(m))
    if r:
        raise ValueError(f"orc_march_seed failed ({r}) for {eq!r}")
    out = Mesh()
    for k in ("n1", "n_cells", "n_active", "n_tris", "n_amb", "n_flipped", "fnv_soup"):
        setattr(out, k, getattr(m, k))
    nt = m.n_tris
    out.soup = (np.ctypeslib.as_array(m.soup, shape=(nt * 9,)).copy().reshape(nt, 3, 3) if (want & WANT_SOUP) and nt
                else np.zeros((0, 3, 3), np.float32))
    out.normals = (np.ctypeslib.as_array(m.normals, shape=(nt * 9,)).copy().reshape(nt, 3, 3)
                   if (want & WANT_NORMALS) and nt else None)
    L.orc_mesh_free(C.byref(m))
    return out


def evaluate(eq: str, x, y, z, pow_mode=POW_LIBM):
    """f(x,y,z) under the reference's evaluation order; None if rejected / stack underflow."""
    e = _Expr()
    if not lib().orc_tokenize(eq.encode(), C.byref(e)):
        return None
    out = C.c_float()
    r = lib().orc_evaluate(C.byref(e), x, y, z, pow_mode, C.byref(out))
    lib().orc_expr_free(C.byref(e))
    return None if r else out.value


def evaluate_many(eq: str, pts: np.ndarray, pow_mode=POW_EXACT) -> np.ndarray:
    e = _Expr()
    if not lib().orc_tokenize(eq.encode(), C.byref(e)):
        raise ValueError("rejected: " + eq)
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
    res = np.empty(len(pts), dtype=np.float32)
    out = C.c_float()
    L = lib()
    for i, (x, y, z) in enumerate(pts):
        if L.orc_evaluate(C.byref(e), float(x), float(y), float(z), pow_mode, C.byref(out)):
            L.orc_expr_free(C.byref(e))
            raise ValueError("stack underflow: " + eq)
        res[i] = out.value
    L.orc_expr_free(C.byref(e))
    return res


def cells_per_axis(step: float) -> int:
    return lib().orc_cells_per_axis(C.c_float(step))


def axis_coords(step: float) -> np.ndarray:
    n1 = cells_per_axis(step)
    a = np.empty(n1 + 1, dtype=np.float32)
    lib().orc_axis_coords(C.c_float(step), a.ctypes.data_as(C.POINTER(C.c_float)), n1 + 1)
    return a


def fnv1a(buf: bytes, h=1469598103934665603) -> int:
    return lib().orc_fnv1a(buf, len(buf), h)


class Mesh:
    pass


def set_extensions(ext: int) -> int:
    """Grammar extension E1 (sin/cos), not part of the reference; returns the previous setting."""
    lib().orc_set_extensions.argtypes = [C.c_uint]
    lib().orc_set_extensions.restype = C.c_uint
    return lib().orc_set_extensions(ext)


CMP = {">=": 0, "<=": 1, ">": 2, "<": 3}  # ORC_CMP_*; the strings marching.cpp:181-190 accepts


def march(eq: str, step: float, iso: float = 0.0, scale=(1.0, 1.0, 1.0), pow_mode=POW_LIBM,
          want=WANT_CODES | WANT_SOUP, z_begin=0, z_end=-1, nthreads=None, constraints=()) -> Mesh:
    """Run the oracle sweep.  Returns counts, fingerprints and (copied) numpy arrays.
    constraints: up to three (lhs, op, rhs) with op in '>=', '<=', '>', '<' (marching.cpp:173-200)."""
    if nthreads is None:
        nthreads = os.cpu_count() or 1
    m = _Mesh()
    sc = (C.c_float * 3)(*scale)
    cons = (_Constraint * max(len(constraints), 1))()
    for i, (lhs, op, rhs) in enumerate(constraints):
        cons[i] = _Constraint(lhs.encode(), CMP[op], float(rhs))
    r = lib().orc_march_constrained(eq.encode(), C.c_float(step), C.c_float(iso), sc, pow_mode, want, z_begin, z_end,
                                    nthreads, cons, len(constraints), C.byref(m))
    if r:
        raise ValueError(f"orc_march failed ({r}) for {eq!r}")
    out = Mesh()
    for k in ("n1", "n_cells", "n_active", "n_tris", "n_amb", "n_flipped", "fnv_codes", "fnv_soup"):
        setattr(out, k, getattr(m, k))
    out.codes = np.ctypeslib.as_array(m.codes, shape=(m.n_cells,)).copy() if (want & WANT_CODES) else None
    nt = m.n_tris
    out.soup = (np.ctypeslib.as_array(m.soup, shape=(nt * 9,)).copy().reshape(nt, 3, 3)
                if (want & WANT_SOUP) and nt else np.zeros((0, 3, 3), np.float32) if (want & WANT_SOUP) else None)
    out.normals = (np.ctypeslib.as_array(m.normals, shape=(nt * 9,)).copy().reshape(nt, 3, 3)
                   if (want & WANT_NORMALS) and nt else np.zeros((0, 3, 3), np.float32) if (want & WANT_NORMALS) else None)
    lib().orc_mesh_free(C.byref(m))
    return out


class _Indexed(C.Structure):
    _fields_ = [("n_verts", C.c_uint64), ("n_tris", C.c_uint64), ("vertex_list", C.POINTER(C.c_float)),
                ("tri_list", C.POINTER(C.c_uint32)), ("normals", C.POINTER(C.c_float))]


def march_indexed(eq: str, step: float, iso: float = 0.0, scale=(1.0, 1.0, 1.0), pow_mode=POW_LIBM, constraints=(),
                  z_begin=0, z_end=-1) -> Mesh:
    """The reference's indexed Poly_Data (marching.cpp:599-654: std::set welding in sweep order) plus the drawer's
    CalculateNormal (normal.h).  Returns .vertices (V,3) f32, .tris (T,3) u32, .normals (V,3) f32."""
    L = lib()
    L.orc_march_indexed.argtypes = [C.c_char_p, C.c_float, C.c_float, C.POINTER(C.c_float), C.c_int, C.POINTER(_Constraint),
                                    C.c_int, C.c_int, C.c_int, C.POINTER(_Indexed)]
    L.orc_march_indexed.restype = C.c_int
    L.orc_indexed_free.argtypes = [C.POINTER(_Indexed)]
    cons = (_Constraint * max(len(constraints), 1))()
    for i, (lhs, op, rhs) in enumerate(constraints):
        cons[i] = _Constraint(lhs.encode(), CMP[op], float(rhs))
    m = _Indexed()
    r = L.orc_march_indexed(eq.encode(), C.c_float(step), C.c_float(iso), (C.c_float * 3)(*scale), pow_mode, cons,
                            len(constraints), z_begin, z_end, C.byref(m))
    if r:
        raise ValueError(f"orc_march_indexed failed ({r}) for {eq!r}")
    out = Mesh()
    nv, nt = m.n_verts, m.n_tris
    out.n_verts, out.n_tris = nv, nt
    out.vertices = np.ctypeslib.as_array(m.vertex_list, shape=(nv * 3,)).copy().reshape(nv, 3) if nv else np.zeros((0, 3), np.float32)
    out.tris = np.ctypeslib.as_array(m.tri_list, shape=(nt * 3,)).copy().reshape(nt, 3) if nt else np.zeros((0, 3), np.uint32)
    out.normals = np.ctypeslib.as_array(m.normals, shape=(nv * 3,)).copy().reshape(nv, 3) if nv else np.zeros((0, 3), np.float32)
    L.orc_indexed_free(C.byref(m))
    return out
