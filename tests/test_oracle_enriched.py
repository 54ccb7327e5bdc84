"""CPU checks of the config-5 (SGFEM-enriched) restatement in oracle/.  No reference arithmetic exists for this
configuration: what can be pinned is (a) that it reduces to the S3 element (config 4) when nothing
is enriched, and (b) the properties any correct enriched element has (symmetry, rigid translations in the kernel,
partition-of-unity reproduction).  -m "not gpu"."""
import numpy as np
import pytest


def _hex(seed=3, distort=0.15):
    rng = np.random.default_rng(seed)
    ref = np.array([(0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1), (1, 0, 1)], float)
    return (0.3 * ref + distort * 0.3 * rng.uniform(-1, 1, size=(8, 3))).reshape(-1)


def _psi_quadratic(x, y, z):
    return (x * x + 0.5 * y * z + 0.25 * z, 2 * x, 0.5 * z, 0.5 * y + 0.25)


def _psi_linear(x, y, z):
    return (0.3 + 1.5 * x - 0.7 * y + 0.4 * z, 1.5 + 0 * x, -0.7 + 0 * x, 0.4 + 0 * x)


def _one_element_record(O, coords, n_enr, funcs, base=408, etype=None, nenrich=None, seed=5):
    rng = np.random.default_rng(seed)
    rec = O.props_enriched(1, 1, 1, _grid_from_hex(coords), n_enr, funcs, base=base,
                           etype=etype, nenrich=nenrich).reshape(-1)
    rec[96:104] = rng.uniform(50, 150, 8)     # per-Gauss-point material
    rec[104:112] = rng.uniform(0.1, 0.4, 8)
    return rec


def _grid_from_hex(coords):
    """the 2x2x2 node grid (natural order) whose single element has the given local-node coordinates"""
    HD = [(0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1), (1, 0, 1)]
    g = np.zeros((2, 2, 2, 3))
    for a, (di, dj, dk) in enumerate(HD):
        g[dk, dj, di] = coords[3 * a:3 * a + 3]
    return g.reshape(-1)


@pytest.mark.parametrize("base", [408, 416])
def test_no_enrichment_is_the_s3_element(O, base):
    coords = _hex()
    rec = _one_element_record(O, coords, 0, [], base=base)
    Ke = O.enr_ke(coords, rec, 0, base=base)
    want = O.ke("s3", coords, rec[96:104], rec[104:112]).reshape(24, 24)
    assert np.array_equal(Ke, want)                                   # same loops, same order: same bits
    L = O.lib()
    Fe, Ge = np.zeros(24), np.zeros(24)
    import ctypes as C
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    L.orc_fe_momentum_hex(dp(coords), dp(rec[160:168].copy()), dp(rec[168:176].copy()), dp(rec[176:184].copy()), dp(Fe))
    L.orc_ge_naturalbc_hex(dp(coords), dp(rec[184:256].copy()), dp(rec[256:328].copy()), dp(rec[328:400].copy()),
                           dp(rec[112:136].copy()), dp(rec[136:160].copy()), dp(Ge))
    F = O.enr_fe(coords, rec, 0, base=base)
    assert np.allclose(F, Fe + Ge, rtol=1e-14, atol=1e-14 * np.abs(Fe + Ge).max())


@pytest.mark.parametrize("n_enr,sgfem", [(1, 1), (2, 1), (1, 0), (3, 0)])
def test_enriched_element_properties(O, n_enr, sgfem):
    coords = _hex(seed=7)
    funcs = [_psi_quadratic, lambda x, y, z: (np.sin(2 * x) * y, 2 * np.cos(2 * x) * y, np.sin(2 * x), 0 * x),
             lambda x, y, z: (z * z * x, z * z, 0 * x, 2 * z * x)][:n_enr]
    rec = _one_element_record(O, coords, n_enr, funcs)
    D = 3 * (1 + n_enr)
    Ke = O.enr_ke(coords, rec, n_enr, sgfem=sgfem)
    scale = np.abs(Ke).max()
    assert np.abs(Ke - Ke.T).max() <= 1e-13 * scale                   # symmetric
    # the standard 24x24 sub-block is the S3 element matrix
    std = np.array([a * D + c for a in range(8) for c in range(3)])
    want = O.ke("s3", coords, rec[96:104], rec[104:112]).reshape(24, 24)
    assert np.abs(Ke[np.ix_(std, std)] - want).max() <= 1e-13 * scale
    # rigid translations of the displacement dofs (enriched dofs zero) produce no force in ANY row
    for c in range(3):
        t = np.zeros(8 * D)
        t[std[c::3]] = 1.0
        assert np.abs(Ke @ t).max() <= 1e-12 * scale
    assert np.linalg.matrix_rank(Ke, tol=1e-9 * scale) <= 8 * D - 6


def test_gfem_reproduces_a_field_the_standard_basis_holds(O):
    """type 0 (GFEM), every node active, psi linear: sum_a N_a psi is trilinear-representable, so setting all enriched
    dofs of component c to 1 gives the same displacement as the standard dofs psi(x_a): identical forces."""
    coords = _hex(seed=9, distort=0.0)       # parallelepiped: N_a psi stays in the Q1 space only if psi*N is... use a box
    rec = _one_element_record(O, coords, 1, [_psi_linear])
    D = 6
    Ke = O.enr_ke(coords, rec, 1, sgfem=0)
    # u = sum_a N_a psi(x) e_c  vs  u = sum_a N_a psi(x_a) e_c: equal iff psi is constant on the element, so compare
    # the weaker, exact statement: psi = const (= 2.5) -> enriched dofs act as 2.5 x the standard ones
    rec2 = _one_element_record(O, coords, 1, [lambda x, y, z: (2.5 + 0 * x, 0 * x, 0 * x, 0 * x)])
    Ke2 = O.enr_ke(coords, rec2, 1, sgfem=0)
    for c in range(3):
        e = np.zeros(8 * D); s = np.zeros(8 * D)
        e[[a * D + 3 + c for a in range(8)]] = 1.0
        s[[a * D + c for a in range(8)]] = 2.5
        assert np.abs(Ke2 @ e - Ke2 @ s).max() <= 1e-12 * np.abs(Ke2).max()
    # SGFEM with a psi that IS trilinear on the element: phi = psi - I_h psi = 0 -> enriched rows and columns vanish
    rec3 = _one_element_record(O, _hex(seed=9, distort=0.0), 1, [_psi_linear])
    Ke3 = O.enr_ke(_hex(seed=9, distort=0.0), rec3, 1, sgfem=1)
    enr = np.array([a * D + 3 + c for a in range(8) for c in range(3)])
    assert np.abs(Ke3[enr]).max() <= 1e-12 * np.abs(Ke3).max() and np.abs(Ke3[:, enr]).max() <= 1e-12 * np.abs(Ke3).max()
    assert np.abs(Ke).max() > 0


def test_inactive_families_vanish_and_are_blocked(O):
    M = N = P = 3
    coords = O.perturb_coords(O.uniform_coords(3, M, N, P), 3, M, N, P)
    n_enr = 2
    et = np.zeros((P, N, M), dtype=np.int64)
    ne = np.zeros((P, N, M), dtype=np.int64)
    et[1, 1, 1] = 1; ne[1, 1, 1] = 1            # the centre node: family 0 only
    et[0, 0, 2] = 3; ne[0, 0, 2] = 2            # a corner node: both families
    props = O.props_enriched(2, 2, 2, coords, n_enr, [_psi_quadratic, _psi_linear], etype=et, nenrich=ne)
    D = 9
    rp, ci = O.pattern(3, D, M, N, P)
    K, b = O.assemble_enriched3d(M, N, P, coords, props, rp, ci, n_enr)
    Kd = np.zeros((D * 27, D * 27))
    for r in range(D * 27):
        Kd[r, ci[rp[r]:rp[r + 1]]] = K[rp[r]:rp[r + 1]]
    assert np.abs(Kd - Kd.T).max() <= 1e-12 * np.abs(Kd).max()
    centre, corner = 1 + 3 * (1 + 3 * 1), 2
    for node in range(27):
        for m in range(n_enr):
            rows = [D * node + 3 + 3 * m + c for c in range(3)]
            active = (node == centre and m == 0) or (node == corner)
            assert (np.abs(Kd[rows]).max() > 0) == active
    O.bc_enriched3d(M, N, P, coords, props, rp, ci, K, b, n_enr, dirichlet=False)
    Kd2 = np.zeros_like(Kd)
    for r in range(D * 27):
        Kd2[r, ci[rp[r]:rp[r + 1]]] = K[rp[r]:rp[r + 1]]
    for node in range(27):
        for m in range(n_enr):
            active = (node == centre and m == 0) or (node == corner)
            for c in range(3):
                r = D * node + 3 + 3 * m + c
                if not active:
                    want = np.zeros(D * 27); want[r] = 1.0
                    assert np.array_equal(Kd2[r], want) and b[r] == 0.0
                else:
                    assert np.array_equal(Kd2[r], Kd[r])
