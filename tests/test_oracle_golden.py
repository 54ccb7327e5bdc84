"""CPU tests of the oracle: pinned to the golden values of SURVEY.md 8(c) (outputs of the reference's
own element kernels) and to the reference's only known-answer check (manufactured solution, error
falling O(h^2): P3:524, S3:1142).  PETSc pattern/BC semantics are restated, not pinned."""
import json
import math
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "survey_8c.json")))

HDI = [0, 0, 1, 1, 0, 0, 1, 1]
HDJ = [0, 1, 1, 0, 0, 1, 1, 0]
HDK = [0, 0, 0, 0, 1, 1, 1, 1]


def cube(h):
    return np.array([[HDI[n] * h, HDJ[n] * h, HDK[n] * h] for n in range(8)], dtype=np.float64).reshape(-1)


def seqsum(v):
    s = 0.0
    for x in v:
        s += float(x)
    return s


def test_golden_poisson_element(O):
    g = GOLD["poisson3d_h_1_32"]
    Ke = O.ke("p3", cube(g["h"]))
    assert Ke[0] == g["Ke0"] and Ke[1] == g["Ke1"] and Ke[6] == g["Ke6"]
    assert abs(seqsum(Ke[:8]) - g["rowsum0"]) < 1e-19
    Fe = O.fe_poisson(cube(g["h"]))
    assert Fe[0] == g["Fe0"] and Fe[6] == g["Fe6"]
    K = Ke.reshape(8, 8)
    assert np.array_equal(K, K.T)  # exactly symmetric: (g_ki*g_kj) commutes


def test_golden_elastic3d_element(O):
    g = GOLD["elastic3d_h_1_128_E90_nu028"]
    Ke = O.ke("s3", cube(g["h"]), [90.0] * 8, [0.28] * 8)
    assert Ke[0] == g["Ke0"] and Ke[1] == g["Ke1"] and Ke[3] == g["Ke3"]
    assert seqsum(Ke[25 * i] for i in range(24)) == g["trace"]
    assert math.sqrt(seqsum(float(v) * float(v) for v in Ke)) == g["frobenius"]
    K = Ke.reshape(24, 24)
    assert 0 < np.abs(K - K.T).max() <= 1.5e-17


def test_golden_planestrain_element(O):
    g = GOLD["planestrain2d_h_1_2048_E1_nu033"]
    h = g["h"]
    q = np.array([0, 0, 0, h, h, h, h, 0.0])
    Ke = O.ke("e2", q, [1.0] * 4, [0.33] * 4)
    assert Ke[0] == g["Ke0"] and Ke[1] == g["Ke1"] and Ke[2] == g["Ke2"]
    assert abs(math.sqrt(seqsum(float(v) * float(v) for v in Ke)) - g["frobenius"]) < 1e-15


@pytest.mark.parametrize("nsd,dof,dims", [(3, 1, (4, 5, 3)), (2, 2, (6, 4, 1)), (3, 3, (3, 3, 4)), (3, 3, (2, 2, 2))])
def test_pattern_is_dmda_box_stencil(O, nsd, dof, dims):
    M, N, P = dims
    rp, ci = O.pattern(nsd, dof, M, N, P)
    nn = M * N * P
    assert len(rp) == dof * nn + 1 and rp[0] == 0
    # closed-form count: dof^2 * prod(3M-2)
    tot = dof * dof * (3 * M - 2) * (3 * N - 2) * ((3 * P - 2) if nsd == 3 else 1)
    assert rp[-1] == tot == len(ci)
    # independent construction: node adjacency = Chebyshev distance <= 1
    idx = np.arange(nn)
    i, j, k = idx % M, (idx // M) % N, idx // (M * N)
    for n in range(nn):
        near = np.nonzero((abs(i - i[n]) <= 1) & (abs(j - j[n]) <= 1) & (abs(k - k[n]) <= 1))[0]
        cols = (dof * near[:, None] + np.arange(dof)[None, :]).reshape(-1)
        for c in range(dof):
            r = dof * n + c
            got = ci[rp[r]:rp[r + 1]]
            assert np.array_equal(got, cols)  # sorted ascending, complete


def test_uniform_coords_is_min_plus_i_h(O):
    M, N, P = 5, 4, 3
    c = O.uniform_coords(3, M, N, P).reshape(P, N, M, 3)
    hx = 1.0 / (M - 1)
    assert c[2, 3, 4, 0] == 0.0 + hx * 4 and c[0, 0, 0, 0] == 0.0
    assert c[1, 2, 3, 1] == (1.0 / (N - 1)) * 2 and c[2, 0, 0, 2] == (1.0 / (P - 1)) * 2


def _csr(rp, ci, v, n):
    return sp.csr_matrix((v, ci.astype(np.int64), rp), shape=(n, n))


def test_poisson_known_answer_converges_second_order(O):
    """P3's own check: L2 error of the FE solution against sin(pi x) sin(pi y) sin(pi z) must fall ~4x per
    halving of h (P3:416-532).  Pins K, b, the in-loop Dirichlet rows and the pattern end to end."""
    errs = []
    for nel in (4, 8, 16):
        M = nel + 1
        coords = O.uniform_coords(3, M, M, M)
        rp, ci = O.pattern(3, 1, M, M, M)
        K, b = O.assemble_poisson3d(M, M, M, coords, rp, ci)
        x = spla.spsolve(_csr(rp, ci, K, M ** 3).tocsc(), b)
        h, l2, h1 = O.errors("p3", M, M, M, coords, x)
        assert h == 1.0 / nel
        errs.append(l2)
    assert 3.5 < errs[0] / errs[1] < 4.6 and 3.7 < errs[1] / errs[2] < 4.3
    assert errs[2] < 3e-3


def test_elastic3d_known_answer_converges_second_order(O):
    """S3's check (S3:1041-1151) with the natural-BC term and all-face Dirichlet rows."""
    errs = []
    for nel in (4, 8):
        M = nel + 1
        coords = O.uniform_coords(3, M, M, M)
        props = O.props_s3(nel, nel, nel, coords)
        rp, ci = O.pattern(3, 3, M, M, M)
        K, b = O.assemble_elastic3d(M, M, M, coords, props, rp, ci)
        O.bc_s3(M, M, M, coords, rp, ci, K, b)
        x = spla.spsolve(_csr(rp, ci, K, 3 * M ** 3).tocsc(), b)
        h, l2, h1 = O.errors("s3", M, M, M, coords, x)
        errs.append(l2)
    assert 3.2 < errs[0] / errs[1] < 4.8


def test_poisson_boundary_rows_are_valence_times_identity(O):
    M = 4
    coords = O.uniform_coords(3, M, M, M)
    rp, ci = O.pattern(3, 1, M, M, M)
    K, b = O.assemble_poisson3d(M, M, M, coords, rp, ci)
    A = _csr(rp, ci, K, M ** 3).toarray()
    for n in range(M ** 3):
        i, j, k = n % M, (n // M) % M, n // (M * M)
        onb = [i in (0, M - 1), j in (0, M - 1), k in (0, M - 1)]
        if any(onb):
            val = 1
            for d, x in zip(onb, (i, j, k)):
                val *= 1 if d else 2
            row = A[n].copy()
            assert row[n] == val and b[n] == 0.0
            row[n] = 0
            assert not row.any()


def test_elastic_rigid_translation_is_in_the_kernel_before_bcs(O):
    M, N, P = 4, 5, 3
    coords = O.perturb_coords(O.uniform_coords(3, M, N, P), 3, M, N, P)
    props = O.props_s3(M - 1, N - 1, P - 1, coords)
    rp, ci = O.pattern(3, 3, M, N, P)
    K, _ = O.assemble_elastic3d(M, N, P, coords, props, rp, ci, want_b=False)
    A = _csr(rp, ci, K, 3 * M * N * P)
    for c in range(3):
        t = np.zeros(3 * M * N * P)
        t[c::3] = 1.0
        assert np.abs(A @ t).max() < 1e-11 * np.abs(K).max()
    assert abs(A - A.T).max() < 1e-13 * np.abs(K).max()


def test_zero_rows_compression_matches_petsc_default(O):
    M, N = 5, 4
    coords = O.uniform_coords(2, M, N)
    props = O.props_e2(M - 1, N - 1, coords)
    rp, ci = O.pattern(2, 2, M, N)
    K, b = O.assemble_elastic2d(M, N, coords, props, rp, ci)
    O.bc_e2(M, N, rp, ci, K, b)
    flags = O.bc_flags("e2", M, N)
    rp2, ci2, K2 = O.compress_zeroed_rows(flags, rp, ci, K)
    A, A2 = _csr(rp, ci, K, 2 * M * N), _csr(rp2, ci2, K2, 2 * M * N)
    assert abs(A - A2).max() == 0.0
    for r in np.nonzero(flags)[0]:
        assert rp2[r + 1] - rp2[r] == 1 and ci2[rp2[r]] == r and K2[rp2[r]] == 1.0
    # E2:882-885
    assert b[2 * (M - 1)] == -1.0 and b[0] == 1.0 and b[1] == 0.0


def test_slab_element_ranges_sum_to_the_global_assembly(O):
    """Reference ownership: rank r assembles element layers [ks-1, ke-1); the contributions of all
    ranks add up to the serial matrix (MatAssemblyEnd stash semantics)."""
    from petsc_amd import slab
    M, N, P = 4, 3, 8
    coords = O.perturb_coords(O.uniform_coords(3, M, N, P), 3, M, N, P)
    rp, ci = O.pattern(3, 1, M, N, P)
    K, b = O.assemble_poisson3d(M, N, P, coords, rp, ci)
    Ks, bs = np.zeros_like(K), np.zeros_like(b)
    for ks, ke in slab.slab_ranges(P, 3):
        e0, e1 = slab.element_range(ks, ke)
        k1, b1 = O.assemble_poisson3d(M, N, P, coords, rp, ci, e0=e0, e1=e1)
        Ks += k1
        bs += b1
    assert np.abs(Ks - K).max() < 1e-15 and np.abs(bs - b).max() < 1e-15


def test_oracle_reproduces_committed_system_fixtures(O):
    """tests/golden/systems.npz (small full systems incl. distorted meshes, per-Gauss-point material, natural-BC
    term, R3 stride) was written by this oracle; it must keep reproducing it to the last bit."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_system_fixtures",
                                                  os.path.join(HERE, "golden", "make_system_fixtures.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    fix = np.load(os.path.join(HERE, "golden", "systems.npz"))
    cases = gen.cases()
    assert len(cases) == 11
    for name, d in cases.items():
        for k in ("coords", "props", "K", "b"):
            assert np.array_equal(fix[name + "/" + k], d[k]), (name, k)


def test_restatement_is_clean_under_asan_and_ubsan():
    """the CPU restatement under -fsanitize=address,undefined (host code only).  oracle/san_check.c
    drives every assembly routine (P3, E2, S3, the enriched kind with both PetscInt widths) on small distorted meshes
    and re-checks two SURVEY 8(c) golden values inside the instrumented build; a sanitizer report aborts the run."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "san_check"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([os.path.join(root, "oracle", "san_check")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "san_check: ok" in r.stdout
