"""Host-side binding of the femasm C ABI (include/femasm.h) for Python callers.

This is plumbing only: the product is libfemasm.so (hand-written gfx950 kernels behind a
plain C ABI).  The binding mirrors the reference drivers' call sites --
``AssembleA_* + AssembleF_* (+ DMDABCApplyCompression)`` become one
``Plan.assemble_system`` call (P3:675-678, E2:692-700, S3:1384-1392).

There is NO CPU path: if libfemasm.so is missing, or no HIP device is present, every
compute entry point raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FEMASM_LIB", os.path.join(_HERE, "csrc", "libfemasm.so"))  # override: kernel experiments

POISSON3D_Q1, PLANE_STRAIN2D_Q1, ELASTIC3D_Q1, ELASTIC3D_ENRICHED = 0, 1, 2, 3
MAT_AOS, MAT_UNIFORM, MAT_MANUFACTURED, MAT_ELEMENT_SOA = 0, 1, 2, 3
BC_NONE, BC_POISSON_INLOOP, BC_E2_COMPRESSION, BC_S3_ALLFACES_EXACT = 0, 1, 2, 3

_TRAITS = {  # kind: (nsd, dof, nen, ngp)
    POISSON3D_Q1: (3, 1, 8, 8),
    PLANE_STRAIN2D_Q1: (2, 2, 4, 4),
    ELASTIC3D_Q1: (3, 3, 8, 8),
    ELASTIC3D_ENRICHED: (3, 3, 8, 8),  # dof = 3 (1 + n_enr), set per plan
}

EXPORTS = [
    "femasm_plan_create", "femasm_plan_destroy", "femasm_plan_set_stream", "femasm_pattern",
    "femasm_pattern_copy", "femasm_scatter_map", "femasm_assemble_system", "femasm_interface_buffers",
    "femasm_interface_set_buffers", "femasm_interface_apply", "femasm_compress_bc_rows", "femasm_plan_sync", "femasm_last_error",
    "femasm_version", "femasm_plan_set_fast_path", "femasm_plan_get_fast_path", "femasm_assemble_interface", "femasm_error_integrals", "femasm_fill_properties", "femasm_apply_operator",
    "femasm_coo_indices", "femasm_coord_moments", "femasm_rigid_body_modes", "femasm_zero_rows",
    "femasm_error_integrals_static", "femasm_plan_set_enriched_contraction", "femasm_interface_exchange", "femasm_plan_set_async", "femasm_eliminate_columns", "femasm_plan_set_option",
]


class FemasmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"femasm error {code}: {msg}")
        self.code = code


class Mesh(C.Structure):
    _fields_ = [("problem", C.c_int32), ("n_enr", C.c_int32), ("M", C.c_int64), ("N", C.c_int64),
                ("P", C.c_int64), ("ks", C.c_int64), ("ke", C.c_int64), ("cks", C.c_int64), ("cke", C.c_int64),
                ("coords", C.c_void_p)]


class Material(C.Structure):
    _fields_ = [("layout", C.c_int32), ("stride", C.c_int32), ("e_ks", C.c_int64), ("gp", C.c_void_p),
                ("E", C.c_double), ("nu", C.c_double), ("enr_mode", C.c_int32), ("reserved", C.c_int32)]


class Quad(C.Structure):
    _fields_ = [("ngp", C.c_int32), ("rule", C.c_int32)]


class BC(C.Structure):
    _fields_ = [("kind", C.c_int32), ("symmetric", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("ms_total", C.c_double), ("ms_kernel_K", C.c_double), ("ms_kernel_rhs", C.c_double),
                ("ms_staging", C.c_double), ("bytes_K", C.c_double), ("bytes_system", C.c_double),
                ("nel", C.c_int64), ("nrows", C.c_int64), ("nnz", C.c_int64), ("bad_detJ", C.c_int64),
                ("ms_kernel_main", C.c_double), ("bytes_main", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def load():
    """Load libfemasm.so; raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; "
                                    "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        try:
            # torch bundles a HIP runtime of its own (libamdhip64.so.7): it has to be in the process before
            # libfemasm.so pulls in /opt/rocm's, or the process ends up with two runtimes and this library's one
            # sees no device (found with `python __graft_entry__.py smoke`, where build() loaded the library first)
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        lib.femasm_last_error.restype = C.c_char_p
        lib.femasm_version.restype = C.c_char_p
        for name in EXPORTS:
            getattr(lib, name)  # AttributeError if a declared symbol is not exported
        _lib = lib
    return _lib


def _ptr(x):
    """Address of a numpy array (host) or a torch tensor (host or device); None -> NULL."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"]
        return C.c_void_p(x.ctypes.data)
    if hasattr(x, "data_ptr"):
        assert x.is_contiguous()
        return C.c_void_p(x.data_ptr())
    if isinstance(x, int):
        return C.c_void_p(x)
    raise TypeError(type(x))


def _check(rc):
    if rc != 0:
        raise FemasmError(rc, load().femasm_last_error().decode())


class Plan:
    """One slab of the structured (DMDA) mesh on one GPU."""

    def __init__(self, problem, M, N, P=1, coords=None, ks=0, ke=None, cks=None, cke=None, device=0, n_enr=0,
                 options=None):
        lib = load()
        nsd, dof, nen, ngp = _TRAITS[problem]
        if problem == ELASTIC3D_ENRICHED:
            dof = 3 * (1 + n_enr)
        self.n_enr = n_enr
        SL = P if nsd == 3 else N
        ke = SL if ke is None else ke
        cks = (max(ks - 1, 0)) if cks is None else cks
        cke = (min(ke + 1, SL)) if cke is None else cke
        self.problem, self.M, self.N, self.P = problem, M, N, (P if nsd == 3 else 1)
        self.nsd, self.dof, self.nen, self.ngp = nsd, dof, nen, ngp
        self.ks, self.ke, self.cks, self.cke = ks, ke, cks, cke
        self._coords = coords  # keep alive
        mesh = Mesh(problem, n_enr, M, N, self.P, ks, ke, cks, cke, _ptr(coords))
        quad = Quad(ngp, 0)
        h = C.c_void_p()
        _check(lib.femasm_plan_create(C.byref(mesh), C.byref(quad), C.c_int(device), C.byref(h)))
        self._h = h
        nrows, nnz = C.c_int64(), C.c_int64()
        rp, ci = C.c_void_p(), C.c_void_p()
        _check(lib.femasm_pattern(self._h, C.byref(nrows), C.byref(nnz), C.byref(rp), C.byref(ci)))
        self.nrows, self.nnz = nrows.value, nnz.value
        self.rowptr_dev, self.colidx_dev = rp.value, ci.value
        plane_e = (M - 1) * (N - 1) if nsd == 3 else (M - 1)
        self.els = max(ks - 1, 0)
        self.ele = ke - 1
        self.nel = plane_e * max(self.ele - self.els, 0)
        for name, value in (options or {}).items():
            self.set_option(name, value)

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:  # _lib is gone at interpreter shutdown
            _lib.femasm_plan_destroy(self._h)
            self._h = None

    __del__ = close

    def set_option(self, name, value):
        """Kernel variants / tile parameters for A/B runs and tests (include/femasm.h: femasm_plan_set_option)."""
        _check(load().femasm_plan_set_option(self._h, C.c_char_p(name.encode()), C.c_int64(int(value))))

    def set_stream(self, stream_handle):
        _check(load().femasm_plan_set_stream(self._h, C.c_void_p(stream_handle)))

    def set_fast_path(self, level):
        """2 = box kernel, 1 = affine kernel, 0 = general kernel (see include/femasm.h)."""
        _check(load().femasm_plan_set_fast_path(self._h, C.c_int(int(level))))

    def fast_path(self):
        lv = C.c_int()
        _check(load().femasm_plan_get_fast_path(self._h, C.byref(lv)))
        return lv.value

    def set_async(self, on=True):
        _check(load().femasm_plan_set_async(self._h, C.c_int(1 if on else 0)))

    def sync(self):
        _check(load().femasm_plan_sync(self._h))

    def pattern(self):
        """(rowptr int64 [nrows+1], colidx int32 [nnz]) as host numpy arrays."""
        rowptr = np.empty(self.nrows + 1, dtype=np.int64)
        colidx = np.empty(self.nnz, dtype=np.int32)
        _check(load().femasm_pattern_copy(self._h, _ptr(rowptr), _ptr(colidx)))
        return rowptr, colidx

    def scatter_map(self):
        nr = self.nen * self.dof
        out = np.empty(self.nel * nr * nr, dtype=np.int64)
        _check(load().femasm_scatter_map(self._h, _ptr(out)))
        return out.reshape(self.nel, nr, nr)

    def set_enriched_contraction(self, mode):
        """-1 default (= 2), 0 / 1: element matrices on the FP64 VALU / FP64 MFMA (n_enr = 1) + segmented reduction,
        2: one node-centric pass."""
        _check(load().femasm_plan_set_enriched_contraction(self._h, C.c_int(int(mode))))

    def assemble_system(self, values, b, material=None, bc=BC_NONE, E=0.0, nu=0.0, stride=0, e_ks=None,
                        want_stats=False, enr_mode=0, symmetric=False, manufactured=False, element_soa=False):
        """values / b: numpy (host) or torch (device) float64 buffers, or None.
        material: AoS GaussPointCoefficients array (numpy/torch) or None for uniform E, nu; with element_soa=True a
        device tensor of two planes, E per element then nu per element, `stride` doubles apart (K only)."""
        if material is None:
            mat = Material(MAT_MANUFACTURED if manufactured else MAT_UNIFORM, 0, 0, None, E, nu)
        elif element_soa:
            mat = Material(MAT_ELEMENT_SOA, stride, self.els if e_ks is None else e_ks, _ptr(material), E, nu, 0, 0)
        else:
            mat = Material(MAT_AOS, stride, self.els if e_ks is None else e_ks, _ptr(material), E, nu, enr_mode, 0)
        bcs = BC(bc, 1 if symmetric else 0)
        st = Stats()
        _check(load().femasm_assemble_system(self._h, C.byref(mat), C.byref(bcs), _ptr(values), _ptr(b),
                                             C.byref(st) if want_stats else None))
        return st if want_stats else None

    def assemble_interface(self, material=None, bc=BC_NONE, E=0.0, nu=0.0, stride=0, e_ks=None, element_soa=False):
        """First half of a split call: fill the send buffer only, asynchronously (see include/femasm.h)."""
        if material is None:
            mat = Material(MAT_UNIFORM, 0, 0, None, E, nu)
        else:
            mat = Material(MAT_ELEMENT_SOA if element_soa else MAT_AOS, stride, self.els if e_ks is None else e_ks, _ptr(material), E, nu)
        bcs = BC(bc, 0)
        _check(load().femasm_assemble_interface(self._h, C.byref(mat), C.byref(bcs)))

    def interface_buffers(self):
        sb, rb = C.c_void_p(), C.c_void_p()
        sc, rc = C.c_int64(), C.c_int64()
        _check(load().femasm_interface_buffers(self._h, C.byref(sb), C.byref(sc), C.byref(rb), C.byref(rc)))
        return (sb.value, sc.value), (rb.value, rc.value)

    def interface_set_buffers(self, send, recv):
        self._ifbufs = (send, recv)  # keep alive
        _check(load().femasm_interface_set_buffers(self._h, _ptr(send), _ptr(recv)))

    def interface_apply(self, values, b, bc=BC_NONE):
        bcs = BC(bc, 0)
        _check(load().femasm_interface_apply(self._h, C.byref(bcs), _ptr(values), _ptr(b)))

    def eliminate_columns(self, values, b, bc):
        """opt-in symmetric elimination on a complete rows-only system (device tensors)"""
        bcs = BC(bc, 1)
        _check(load().femasm_eliminate_columns(self._h, C.byref(bcs), _ptr(values), _ptr(b)))

    def apply_operator(self, x, y, material=None, bc=BC_NONE, E=0.0, nu=0.0, stride=0, e_ks=None, element_soa=False):
        """N4: y = K x matrix-free (device tensors; x on the plan's coordinate layers, y the owned rows)."""
        if material is None:
            mat = Material(MAT_UNIFORM, 0, 0, None, E, nu)
        else:
            mat = Material(MAT_ELEMENT_SOA if element_soa else MAT_AOS, stride, self.els if e_ks is None else e_ks, _ptr(material), E, nu)
        bcs = BC(bc, 0)
        _check(load().femasm_apply_operator(self._h, C.byref(mat), C.byref(bcs), _ptr(x), _ptr(y)))

    def fill_properties(self, out, E, nu, stride):
        """N1: the drivers' GaussPointCoefficients fill for this slab's element layers, into `out`."""
        _check(load().femasm_fill_properties(self._h, C.c_double(E), C.c_double(nu), C.c_int32(stride), _ptr(out)))

    def error_integrals(self, u):
        """(sum of L2 integrals, sum of H1 integrals) over this slab's elements; u on the plan's coordinate layers."""
        out = (C.c_double * 2)()
        _check(load().femasm_error_integrals(self._h, _ptr(u), out))
        return out[0], out[1]

    def error_integrals_static(self, u, sums):
        """The no-allocation variant: u and sums (2 doubles) are device tensors; asynchronous on the plan's stream."""
        _check(load().femasm_error_integrals_static(self._h, _ptr(u), _ptr(sums)))

    def coo_indices(self, coo_i, coo_j):
        """Global (row, column) of every CSR value slot, into two int32 or int64 arrays/tensors of nnz entries."""
        nbytes = coo_i.dtype.itemsize if isinstance(coo_i, np.ndarray) else coo_i.element_size()
        _check(load().femasm_coo_indices(self._h, C.c_int32(nbytes), _ptr(coo_i), _ptr(coo_j)))

    def coord_moments(self):
        """[count, Sx, Sy, Sz, Sxx, Syy, Szz, Sxy, Syz, Szx] over the owned nodes (add over ranks)."""
        out = (C.c_double * 10)()
        _check(load().femasm_coord_moments(self._h, out))
        return np.array(out[:], dtype=np.float64)

    def rigid_body_modes(self, moments, out):
        """Orthonormal rigid-body modes (6 x nrows, 3 x nrows in 2-D) from the GLOBAL moments."""
        mom = (C.c_double * 10)(*[float(v) for v in moments])
        _check(load().femasm_rigid_body_modes(self._h, mom, _ptr(out)))

    def zero_rows(self, rows, vals, values, b):
        """MatZeroRows(diag 1) + VecSetValues for a list of global rows (int64 array/tensor, host or device)."""
        n = len(rows) if isinstance(rows, np.ndarray) else rows.numel()
        _check(load().femasm_zero_rows(self._h, C.c_int64(n), _ptr(rows), _ptr(vals), _ptr(values), _ptr(b)))

    def compress_bc_rows(self, values, bc):
        """PETSc's truncated MatZeroRows storage; host numpy in, host numpy out."""
        rp = np.empty(self.nrows + 1, dtype=np.int64)
        ci = np.empty(self.nnz, dtype=np.int32)
        va = np.empty(self.nnz, dtype=np.float64)
        n = C.c_int64()
        bcs = BC(bc, 0)
        _check(load().femasm_compress_bc_rows(self._h, C.byref(bcs), _ptr(values), _ptr(rp), _ptr(ci), _ptr(va),
                                              C.byref(n)))
        return rp, ci[:n.value].copy(), va[:n.value].copy()
