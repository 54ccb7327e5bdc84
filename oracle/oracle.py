"""ctypes loader for the CPU oracle (oracle/femasm_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (petsc_amd/) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libfemasm_oracle.so")

c_dp = C.POINTER(C.c_double)
c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)


def build(force=False):
    src = os.path.join(_HERE, "femasm_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libfemasm_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_dp)


def _i64p(a):
    return a.ctypes.data_as(c_i64p)


def _i32p(a):
    return a.ctypes.data_as(c_i32p)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.orc_dmda_nnz.restype = C.c_int64
        _lib.orc_compress_zeroed_rows.restype = C.c_int64
        _lib.orc_poisson_exact.restype = C.c_double
    return _lib


# --------------------------------------------------------------------------
# thin numpy wrappers
# --------------------------------------------------------------------------
def uniform_coords(nsd, M, N, P=1):
    n = M * N * (P if nsd == 3 else 1)
    out = np.empty(n * nsd, dtype=np.float64)
    lib().orc_dmda_uniform_coords(C.c_int(nsd), C.c_int64(M), C.c_int64(N), C.c_int64(P), _dp(out))
    return out


def perturb_coords(coords, nsd, M, N, P=1, amp=0.2, seed=12345):
    """'General geometry' variant of SURVEY 8(d): interior nodes moved by a smooth deterministic
    offset (amp*h) so no element is a parallelepiped; boundary nodes stay put."""
    c = coords.reshape(-1, nsd).copy()
    if nsd == 2:
        P = 1
    k, j, i = np.meshgrid(np.arange(P), np.arange(N), np.arange(M), indexing="ij")
    interior = (i > 0) & (i < M - 1) & (j > 0) & (j < N - 1)
    if nsd == 3:
        interior &= (k > 0) & (k < P - 1)
    interior = interior.reshape(-1)
    h = [1.0 / (M - 1), 1.0 / (N - 1), 1.0 / max(P - 1, 1)]
    ph = (seed % 97) * 0.01
    x = c.copy()
    for d in range(nsd):
        arg = 7.0 * x[:, (d + 1) % nsd] + 5.0 * x[:, (d + 2) % nsd if nsd == 3 else d] + ph + d
        c[:, d] += np.where(interior, amp * h[d] * np.sin(arg), 0.0)
    return c.reshape(-1)


def pattern(nsd, dof, M, N, P=1):
    L = lib()
    nnz = L.orc_dmda_nnz(C.c_int(nsd), C.c_int(dof), C.c_int64(M), C.c_int64(N), C.c_int64(P))
    nrows = dof * M * N * (P if nsd == 3 else 1)
    rowptr = np.empty(nrows + 1, dtype=np.int64)
    colidx = np.empty(nnz, dtype=np.int32)
    L.orc_dmda_pattern(C.c_int(nsd), C.c_int(dof), C.c_int64(M), C.c_int64(N), C.c_int64(P), _i64p(rowptr),
                       _i32p(colidx))
    return rowptr, colidx


def props_e2(mx, my, coords, E=1.0, nu=0.33):
    out = np.empty(24 * mx * my, dtype=np.float64)
    lib().orc_props_e2_fill(C.c_int64(mx), C.c_int64(my), _dp(coords), C.c_double(E), C.c_double(nu), _dp(out))
    return out


def props_s3(mx, my, mz, coords, E=90.0, nu=0.28, stride=400):
    out = np.empty(stride * mx * my * mz, dtype=np.float64)
    lib().orc_props_s3_fill(C.c_int64(mx), C.c_int64(my), C.c_int64(mz), _dp(coords), C.c_double(E),
                            C.c_double(nu), _dp(out), C.c_int(stride))
    return out


def assemble_poisson3d(M, N, P, coords, rowptr, colidx, e0=0, e1=None, want_K=True, want_b=True):
    e1 = P - 1 if e1 is None else e1
    vals = np.zeros(len(colidx), dtype=np.float64) if want_K else None
    b = np.zeros(M * N * P, dtype=np.float64) if want_b else None
    rc = lib().orc_assemble_poisson3d(C.c_int64(M), C.c_int64(N), C.c_int64(P), _dp(coords), _i64p(rowptr),
                                      _i32p(colidx), _dp(vals), _dp(b), C.c_int64(e0), C.c_int64(e1))
    assert rc == 0
    return vals, b


def assemble_elastic2d(M, N, coords, props, rowptr, colidx, stride=24, e0=0, e1=None, want_K=True, want_b=True):
    e1 = N - 1 if e1 is None else e1
    vals = np.zeros(len(colidx), dtype=np.float64) if want_K else None
    b = np.zeros(2 * M * N, dtype=np.float64) if want_b else None
    rc = lib().orc_assemble_elastic2d(C.c_int64(M), C.c_int64(N), _dp(coords), _dp(props), C.c_int(stride),
                                      _i64p(rowptr), _i32p(colidx), _dp(vals), _dp(b), C.c_int64(e0), C.c_int64(e1))
    assert rc == 0
    return vals, b


def assemble_elastic3d(M, N, P, coords, props, rowptr, colidx, stride=400, e0=0, e1=None, want_K=True,
                       want_b=True):
    e1 = P - 1 if e1 is None else e1
    vals = np.zeros(len(colidx), dtype=np.float64) if want_K else None
    b = np.zeros(3 * M * N * P, dtype=np.float64) if want_b else None
    rc = lib().orc_assemble_elastic3d(C.c_int64(M), C.c_int64(N), C.c_int64(P), _dp(coords), _dp(props),
                                      C.c_int(stride), _i64p(rowptr), _i32p(colidx), _dp(vals), _dp(b),
                                      C.c_int64(e0), C.c_int64(e1))
    assert rc == 0
    return vals, b


def bc_e2(M, N, rowptr, colidx, vals, b):
    lib().orc_bc_e2_compression(C.c_int64(M), C.c_int64(N), _i64p(rowptr), _i32p(colidx), _dp(vals), _dp(b))


def bc_s3(M, N, P, coords, rowptr, colidx, vals, b):
    lib().orc_bc_s3_allfaces(C.c_int64(M), C.c_int64(N), C.c_int64(P), _dp(coords), _i64p(rowptr), _i32p(colidx),
                             _dp(vals), _dp(b))


def bc_flags(kind, M, N, P=1):
    if kind == "e2":
        f = np.zeros(2 * M * N, dtype=np.uint8)
        lib().orc_bc_flags_e2(C.c_int64(M), C.c_int64(N), f.ctypes.data_as(c_u8p))
    else:
        f = np.zeros(3 * M * N * P, dtype=np.uint8)
        lib().orc_bc_flags_s3(C.c_int64(M), C.c_int64(N), C.c_int64(P), f.ctypes.data_as(c_u8p))
    return f


def compress_zeroed_rows(flags, rowptr, colidx, vals):
    nrows = len(rowptr) - 1
    rp = np.empty_like(rowptr)
    ci = np.empty_like(colidx)
    va = np.empty_like(vals)
    nnz = lib().orc_compress_zeroed_rows(C.c_int64(nrows), flags.ctypes.data_as(c_u8p), _i64p(rowptr),
                                         _i32p(colidx), _dp(vals), _i64p(rp), _i32p(ci), _dp(va))
    return rp, ci[:nnz].copy(), va[:nnz].copy()


def errors(kind, M, N, P, coords, x):
    h, l2, h1 = C.c_double(), C.c_double(), C.c_double()
    f = lib().orc_errors_poisson3d if kind == "p3" else lib().orc_errors_elastic3d
    f(C.c_int64(M), C.c_int64(N), C.c_int64(P), _dp(coords), _dp(x), C.byref(h), C.byref(l2), C.byref(h1))
    return h.value, l2.value, h1.value


def ke(kind, coords, E=None, nu=None):
    L = lib()
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    if kind == "p3":
        out = np.zeros(64)
        L.orc_ke_poisson_hex(_dp(coords), _dp(out))
    elif kind == "e2":
        out = np.zeros(64)
        L.orc_ke_planestrain_quad(_dp(coords), _dp(np.asarray(E, dtype=np.float64)),
                                  _dp(np.asarray(nu, dtype=np.float64)), _dp(out))
    else:
        out = np.zeros(576)
        L.orc_ke_elastic_hex(_dp(coords), _dp(np.asarray(E, dtype=np.float64)),
                             _dp(np.asarray(nu, dtype=np.float64)), _dp(out))
    return out


def fe_poisson(coords):
    out = np.zeros(8)
    lib().orc_fe_poisson_hex(_dp(np.ascontiguousarray(coords, dtype=np.float64)), _dp(out))
    return out
