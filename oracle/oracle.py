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


# --------------------------------------------------------------------------
# config 5 (SGFEM-enriched hexahedra): the build's own specification, no reference arithmetic exists
# --------------------------------------------------------------------------
def enr_stride(n_enr, base=408):
    return base + 64 * n_enr


def enr_ke(coords, rec, n_enr, sgfem=1, base=408):
    D = 3 * (1 + n_enr)
    out = np.zeros((8 * D) ** 2)
    lib().orc_enr_ke(_dp(np.ascontiguousarray(coords, dtype=np.float64)), _dp(rec), C.c_int(base), C.c_int(n_enr),
                     C.c_int(sgfem), _dp(out))
    return out.reshape(8 * D, 8 * D)


def enr_fe(coords, rec, n_enr, sgfem=1, base=408):
    D = 3 * (1 + n_enr)
    out = np.zeros(8 * D)
    lib().orc_enr_fe(_dp(np.ascontiguousarray(coords, dtype=np.float64)), _dp(rec), C.c_int(base), C.c_int(n_enr),
                     C.c_int(sgfem), _dp(out))
    return out


def assemble_enriched3d(M, N, P, coords, props, rowptr, colidx, n_enr, sgfem=1, base=408, e0=0, e1=None,
                        want_K=True, want_b=True):
    e1 = P - 1 if e1 is None else e1
    D = 3 * (1 + n_enr)
    vals = np.zeros(len(colidx), dtype=np.float64) if want_K else None
    b = np.zeros(D * M * N * P, dtype=np.float64) if want_b else None
    rc = lib().orc_assemble_enriched3d(C.c_int64(M), C.c_int64(N), C.c_int64(P), _dp(coords), _dp(props),
                                       C.c_int(enr_stride(n_enr, base)), C.c_int(base), C.c_int(n_enr),
                                       C.c_int(sgfem), _i64p(rowptr), _i32p(colidx), _dp(vals), _dp(b),
                                       C.c_int64(e0), C.c_int64(e1))
    assert rc == 0
    return vals, b


def bc_enriched3d(M, N, P, coords, props, rowptr, colidx, vals, b, n_enr, dirichlet=True, base=408):
    lib().orc_bc_enriched3d(C.c_int64(M), C.c_int64(N), C.c_int64(P), _dp(coords), _dp(props),
                            C.c_int(enr_stride(n_enr, base)), C.c_int(base), C.c_int(n_enr),
                            C.c_int(1 if dirichlet else 0), _i64p(rowptr), _i32p(colidx), _dp(vals), _dp(b))


def props_enriched(mx, my, mz, coords, n_enr, psi_funcs, etype=None, nenrich=None, base=408, E=90.0, nu=0.28):
    """R3 records (S3 fill + enrich flags) followed by the enrichment block of each family.
    psi_funcs: list of n_enr callables (x, y, z arrays) -> (psi, dpsi/dx, dpsi/dy, dpsi/dz).
    etype / nenrich: int arrays [P, N, M] per NODE (defaults: every node enriched with all families); every element
    record gets its 8 nodes' flags, so the records of the elements around a node agree."""
    stride = enr_stride(n_enr, base)
    s3 = props_s3(mx, my, mz, coords, E=E, nu=nu, stride=400).reshape(-1, 400)
    nel = mx * my * mz
    out = np.zeros((nel, stride))
    out[:, :400] = s3
    M, N, P = mx + 1, my + 1, mz + 1
    if etype is None:
        etype = np.ones((P, N, M), dtype=np.int64)
    if nenrich is None:
        nenrich = np.full((P, N, M), n_enr, dtype=np.int64)
    c4 = coords.reshape(P, N, M, 3)
    HD = [(0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1), (1, 0, 1)]
    flags = np.zeros((nel, 16), dtype=np.int32 if base == 408 else np.int64)
    xn = []
    for a, (di, dj, dk) in enumerate(HD):
        sl = (slice(dk, dk + mz), slice(dj, dj + my), slice(di, di + mx))
        flags[:, a] = etype[sl].reshape(-1)
        flags[:, 8 + a] = nenrich[sl].reshape(-1)
        xn.append(c4[sl].reshape(-1, 3))
    out[:, 400:base] = flags.view(np.float64).reshape(nel, base - 400)
    gp = s3[:, 0:24].reshape(nel, 8, 3)
    gpb = s3[:, 24:96].reshape(nel, 24, 3)
    for m, f in enumerate(psi_funcs):
        o = base + 64 * m
        for a in range(8):
            out[:, o + a] = f(xn[a][:, 0], xn[a][:, 1], xn[a][:, 2])[0]
        v = f(gp[:, :, 0], gp[:, :, 1], gp[:, :, 2])
        out[:, o + 8:o + 16] = v[0]
        out[:, o + 16:o + 40] = np.stack([v[1], v[2], v[3]], axis=-1).reshape(nel, 24)
        out[:, o + 40:o + 64] = f(gpb[:, :, 0], gpb[:, :, 1], gpb[:, :, 2])[0]
    return out.reshape(-1)
