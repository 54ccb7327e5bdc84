"""The BASELINE.json configurations as stated, and the benchmarked (level-2) kernels at the benchmarked sizes
(-m gpu).  What the oracle cannot assemble in seconds is checked through (a) windows of interior rows against an
oracle assembly of the same sub-box, (b) size-independent properties (translations in the kernel of K, unit /
valence Dirichlet rows, no unwritten slot, bitwise repeat), (c) slab-by-slab assembly against the one-plan
system at 1e-13.

  config 1  FEM_poisson3D 32^3 (33^3 nodes): the whole system against the oracle
  config 2  FEM_elsatic2D 2048^2: tests/test_gpu_fullsize.py (uniform mesh, golden diagonal)
  config 3  FEM_poisson3D 256^3 over 8 slabs by the DMDA rule (257 layers -> 33, 32 x 7)
  config 4  SGFEM_elastic3D 128^3, 1 plan and 8 slabs (129 -> 17, 16 x 7), uniform mesh = what bench.py times
  config 4' 256^3 (north_star's scaling size) on one GPU: window in the far corner (value offsets beyond 2^31)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-13


@pytest.fixture(scope="module")
def env(O):
    import torch
    from petsc_amd import femasm as F
    from petsc_amd import slab, synth
    assert torch.cuda.is_available()
    return dict(torch=torch, F=F, O=O, slab=slab, synth=synth, dev=torch.device("cuda:0"))


def _interior_window(n):
    return [(i, j, k) for k in range(1, n - 1) for j in range(1, n - 1) for i in range(1, n - 1)]


def _node_start(slab, dof, M, N, P, i, j, k):
    pairs = slab._pre(k, P) * slab._tot(N) * slab._tot(M) + slab._span(k, P) * (
        slab._pre(j, N) * slab._tot(M) + slab._span(j, N) * slab._pre(i, M))
    return dof * dof * pairs


def _window_check(env, kind, vals, b, coords, props, M, origin, n=6):
    """rows of the strictly interior nodes of the n^3-node sub-box at `origin` against an oracle assembly of the
    sub-box on the same coordinates (and material); returns (rel K, rel b)"""
    torch, O, slab = env["torch"], env["O"], env["slab"]
    i0, j0, k0 = origin
    dof = 1 if kind == "p3" else 3
    c4 = coords.reshape(M, M, M, 3)[k0:k0 + n, j0:j0 + n, i0:i0 + n].contiguous().cpu().numpy().reshape(-1)
    rps, cis = O.pattern(3, dof, n, n, n)
    if kind == "p3":
        Ks, bs = O.assemble_poisson3d(n, n, n, c4, rps, cis)
    else:
        p4 = props.reshape(M - 1, M - 1, M - 1, 400)[k0:k0 + n - 1, j0:j0 + n - 1, i0:i0 + n - 1]
        Ks, bs = O.assemble_elastic3d(n, n, n, c4, p4.contiguous().cpu().numpy().reshape(-1), rps, cis)
    nK = dK = nb = db = 0.0
    blk = 27 * dof * dof
    for (i, j, k) in _interior_window(n):
        g0 = _node_start(slab, dof, M, M, M, i0 + i, j0 + j, k0 + k)
        g = vals[g0:g0 + blk].cpu().numpy()
        rs = dof * (i + n * (j + n * k))
        s = Ks[rps[rs]:rps[rs + dof]]
        assert g.shape == s.shape
        nK += np.sum((g - s) ** 2); dK += np.sum(s ** 2)
        gn = dof * ((i0 + i) + M * ((j0 + j) + M * (k0 + k)))
        bg = b[gn:gn + dof].cpu().numpy()
        nb += np.sum((bg - bs[rs:rs + dof]) ** 2); db += np.sum(bs[rs:rs + dof] ** 2)
    return np.sqrt(nK / dK), np.sqrt(nb / max(db, 1e-300))


def test_config1_poisson3d_32_whole_system(env):
    """BASELINE config 1: FEM_poisson3D on 32^3 elements (P3:657 -nel 32), K + b with the in-loop Dirichlet rows."""
    torch, F, O, dev = env["torch"], env["F"], env["O"], env["dev"]
    M = 33
    coords = O.uniform_coords(3, M, M, M)
    rp, ci = O.pattern(3, 1, M, M, M)
    K, b = O.assemble_poisson3d(M, M, M, coords, rp, ci)
    for level in (2, 1, 0):
        plan = F.Plan(F.POISSON3D_Q1, M, M, M, coords=torch.from_numpy(coords).to(dev))
        plan.set_fast_path(level)
        grp, gci = plan.pattern()
        assert plan.nnz == 912673 and np.array_equal(grp, rp) and np.array_equal(gci, ci)
        vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
        bb = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
        plan.assemble_system(vals, bb, bc=F.BC_POISSON_INLOOP)
        assert plan.fast_path() == level
        v, g = vals.cpu().numpy(), bb.cpu().numpy()
        assert not np.isnan(v).any() and not np.isnan(g).any()
        assert np.linalg.norm(v - K) <= TOL * np.linalg.norm(K)
        assert np.linalg.norm(g - b) <= TOL * np.linalg.norm(b)
        plan.close()


@pytest.mark.parametrize("kkernel", ["unit", "rowsweep"])
def test_config4_elastic3d_128_uniform_level2_is_what_bench_times(env, monkeypatch, kkernel):
    """The benchmarked kernels at the benchmarked size: k_assemble_s3_unit with a whole x line per wave (chosen only at
    such sizes) -- or, with option "s3_rowsweep", k_assemble_rowsweep<ELASTIC> with 64 node layers per workgroup --,
    k_rhs_elem_elastic_runs<true>, k_bc_rows, on DMDASetUniformCoordinates."""
    torch, F, O, synth, dev = env["torch"], env["F"], env["O"], env["synth"], env["dev"]
    nel = 128
    M = nel + 1
    coords = synth.uniform_coords(3, M, M, M, device=dev)
    plan = F.Plan(F.ELASTIC3D_Q1, M, M, M, coords=coords, options={"s3_rowsweep": 1} if kkernel == "rowsweep" else None)
    props = torch.empty(plan.nel * 400, dtype=torch.float64, device=dev)
    plan.fill_properties(props, 90.0, 0.28, 400)   # as bench.py does
    vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
    b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
    plan.assemble_system(vals, b, material=props, stride=400, bc=F.BC_S3_ALLFACES_EXACT)
    assert plan.fast_path() == 2, "the uniform mesh must stay on the verified fast path"
    assert plan.nnz == 513599625 and plan.nrows == 6440067
    assert not torch.isnan(vals).any() and not torch.isnan(b).any()          # no slot unwritten
    # windows: one in the middle, one touching the last tile row / column / node layer of the grid of workgroups
    for origin in ((40, 50, 60), (M - 7, M - 7, M - 7), (1, 60, 120)):
        rK, rb = _window_check(env, "s3", vals, b, coords, props, M, origin)
        assert rK <= TOL and rb <= TOL, (origin, rK, rb)
    # properties on the whole matrix: translations in the kernel of the non-Dirichlet rows, unit Dirichlet rows
    rp, ci = plan.pattern()
    A = torch.sparse_csr_tensor(torch.from_numpy(rp).to(dev), torch.from_numpy(ci.astype(np.int64)).to(dev), vals,
                                size=(plan.nrows, plan.nrows))
    idx = torch.arange(M ** 3, device=dev)
    i, j, k = idx % M, (idx // M) % M, idx // (M * M)
    bnd = ((i == 0) | (i == M - 1) | (j == 0) | (j == M - 1) | (k == 0) | (k == M - 1)).repeat_interleave(3)
    scale = vals.abs().max()
    for c in range(3):
        t = torch.zeros(plan.nrows, dtype=torch.float64, device=dev)
        t[c::3] = 1.0
        y = A @ t
        assert y[~bnd].abs().max() <= 1e-11 * scale
        assert torch.equal(y[bnd], t[bnd])
    del A
    # bitwise repeat
    vals2 = torch.empty_like(vals)
    b2 = torch.empty_like(b)
    plan.assemble_system(vals2, b2, material=props, stride=400, bc=F.BC_S3_ALLFACES_EXACT)
    assert torch.equal(vals, vals2) and torch.equal(b, b2)
    plan.close()


def test_config3_poisson3d_256_uniform_level2_and_8_slabs(env):
    """P3 256^3 on the uniform mesh: k_assemble_p3_rect + the separable right-hand side at the benchmarked size, then
    BASELINE config 3: the same mesh cut into 8 z-slabs by the DMDA rule (257 node layers -> 33, 32 x 7; element
    ownership P3:311-312), one slab after another on this GPU with the send buffers handed down as RCCL would,
    against the one-plan system."""
    torch, F, O, synth, slab, dev = env["torch"], env["F"], env["O"], env["synth"], env["slab"], env["dev"]
    nel = 256
    M = nel + 1
    coords = synth.uniform_coords(3, M, M, M, device=dev)
    plan = F.Plan(F.POISSON3D_Q1, M, M, M, coords=coords)
    vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
    b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
    plan.assemble_system(vals, b, bc=F.BC_POISSON_INLOOP)
    assert plan.fast_path() == 2 and plan.nnz == 454756609
    assert not torch.isnan(vals).any() and not torch.isnan(b).any()
    for origin in ((100, 7, 200), (M - 7, M - 7, M - 7), (1, 1, 1)):
        rK, rb = _window_check(env, "p3", vals, b, coords, None, M, origin)
        assert rK <= TOL and rb <= TOL, (origin, rK, rb)
    # boundary rows = valence * identity, interior row sums = 0, b = 0 on the boundary and > 0 inside
    rp, ci = plan.pattern()
    A = torch.sparse_csr_tensor(torch.from_numpy(rp).to(dev), torch.from_numpy(ci.astype(np.int64)).to(dev), vals,
                                size=(plan.nrows, plan.nrows))
    y = A @ torch.ones(plan.nrows, dtype=torch.float64, device=dev)
    del A
    idx = torch.arange(M ** 3, device=dev)
    i, j, k = idx % M, (idx // M) % M, idx // (M * M)
    f = [(i == 0) | (i == M - 1), (j == 0) | (j == M - 1), (k == 0) | (k == M - 1)]
    bnd = f[0] | f[1] | f[2]
    val = torch.ones_like(y)
    for d in range(3):
        val = val * torch.where(f[d], 1.0, 2.0)
    assert torch.equal(y[bnd], val[bnd])
    assert y[~bnd].abs().max() <= 1e-11 * vals.abs().max()
    assert torch.count_nonzero(b[bnd]) == 0 and (b[~bnd] > 0).all()
    del y, val, idx, i, j, k, f, bnd
    vals2 = torch.empty_like(vals)
    plan.assemble_system(vals2, None, bc=F.BC_POISSON_INLOOP)
    assert torch.equal(vals, vals2)
    del vals2
    plan.close()

    # ---- config 3: 8 slabs ----
    ranges = slab.slab_ranges(M, 8)
    assert [ke - ks for ks, ke in ranges] == [33] + [32] * 7
    _slabs_against_global(env, F.POISSON3D_Q1, 1, M, coords, None, 0, F.BC_POISSON_INLOOP, ranges, vals, b)


def _slabs_against_global(env, kind, dof, M, coords, props, stride, bc, ranges, gvals, gb):
    """one plan per slab, top slab first so that each slab's send buffer exists when the slab below needs it"""
    torch, F, slab, dev = env["torch"], env["F"], env["slab"], env["dev"]
    plane_n, plane_e = M * M, (M - 1) * (M - 1)
    c3 = coords.reshape(M, -1)
    recv_from_above = None
    nK = dK = nb = db = 0.0
    for r in reversed(range(len(ranges))):
        ks, ke = ranges[r]
        cks, cke = max(ks - 1, 0), min(ke + 1, M)
        els, ele = slab.element_range(ks, ke)
        plan = F.Plan(kind, M, M, M, coords=c3[cks:cke].reshape(-1).contiguous(), ks=ks, ke=ke, cks=cks, cke=cke)
        mat = None if props is None else props.reshape(M - 1, -1)[els:ele].reshape(-1).contiguous()
        vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
        b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
        (_, sc), (_, rc) = plan.interface_buffers()
        assert (sc, rc) == slab.interface_counts(3, dof, M, M, M, ks, ke)
        send = torch.full((max(sc, 1),), float("nan"), dtype=torch.float64, device=dev)
        recv = torch.full((max(rc, 1),), float("nan"), dtype=torch.float64, device=dev)
        if rc:
            recv[:rc].copy_(recv_from_above[:rc])          # "RCCL": slab r+1 -> slab r
        plan.interface_set_buffers(send if sc else None, recv if rc else None)
        plan.assemble_interface(material=mat, stride=stride, bc=bc, e_ks=els)   # the overlap path of bench.py
        plan.assemble_system(vals, b, material=mat, stride=stride, bc=bc, e_ks=els)
        plan.interface_apply(vals, b, bc=bc)
        assert plan.fast_path() == 2
        assert not torch.isnan(vals).any() and not torch.isnan(b).any()
        lo = slab.layer_nnz(3, dof, M, M, M, 0, ks)
        gK = gvals[lo:lo + plan.nnz]
        nK += float(torch.sum((vals - gK) ** 2)); dK += float(torch.sum(gK ** 2))
        gbs = gb[plane_n * dof * ks:plane_n * dof * ke]
        nb += float(torch.sum((b - gbs) ** 2)); db += float(torch.sum(gbs ** 2))
        recv_from_above = send
        plan.close()
        del vals, b, recv
    assert np.sqrt(nK / dK) <= TOL and np.sqrt(nb / db) <= TOL, (np.sqrt(nK / dK), np.sqrt(nb / db))


def test_config4_elastic3d_128_over_8_slabs(env):
    """BASELINE config 4 on 8 GPUs, rehearsed on one: 129 node layers -> 17, 16 x 7 (S3:685-686 ownership)."""
    torch, F, synth, slab, dev = env["torch"], env["F"], env["synth"], env["slab"], env["dev"]
    nel = 128
    M = nel + 1
    coords = synth.uniform_coords(3, M, M, M, device=dev)
    plan = F.Plan(F.ELASTIC3D_Q1, M, M, M, coords=coords)
    props = torch.empty(plan.nel * 400, dtype=torch.float64, device=dev)
    plan.fill_properties(props, 90.0, 0.28, 400)
    gvals = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
    gb = torch.empty(plan.nrows, dtype=torch.float64, device=dev)
    plan.assemble_system(gvals, gb, material=props, stride=400, bc=F.BC_S3_ALLFACES_EXACT)
    plan.close()
    ranges = slab.slab_ranges(M, 8)
    assert [ke - ks for ks, ke in ranges] == [17] + [16] * 7
    _slabs_against_global(env, F.ELASTIC3D_Q1, 3, M, coords, props, 400, F.BC_S3_ALLFACES_EXACT, ranges, gvals, gb)


def test_north_star_256_elastic_on_one_gpu_far_corner_window(env):
    """S3 256^3 (50.9 M rows, 4.09e9 values = 32.7 GB: value offsets beyond 2^31) on the uniform mesh: no slot
    unwritten, a window in the far corner against the oracle."""
    torch, F, synth, slab, dev = env["torch"], env["F"], env["synth"], env["slab"], env["dev"]
    free, _ = torch.cuda.mem_get_info()
    if free < 120e9:
        pytest.skip("needs ~100 GB of HBM")
    nel = 256
    M = nel + 1
    coords = synth.uniform_coords(3, M, M, M, device=dev)
    plan = F.Plan(F.ELASTIC3D_Q1, M, M, M, coords=coords)
    assert plan.nnz == 4092809481
    props = torch.empty(plan.nel * 400, dtype=torch.float64, device=dev)
    plan.fill_properties(props, 90.0, 0.28, 400)
    vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
    b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
    plan.assemble_system(vals, b, material=props, stride=400, bc=F.BC_S3_ALLFACES_EXACT)
    assert plan.fast_path() == 2
    assert not torch.isnan(b).any()
    step = 1 << 29                                  # the NaN scan in pieces: a 4 GB mask at once is not needed
    for o in range(0, plan.nnz, step):
        assert not torch.isnan(vals[o:o + step]).any()
    assert _node_start(slab, 3, M, M, M, M - 2, M - 2, M - 2) > 2 ** 31    # value offsets overflow int32
    rK, rb = _window_check(env, "s3", vals, b, coords, props, M, (M - 7, M - 7, M - 7))
    assert rK <= TOL and rb <= TOL, (rK, rb)
    plan.close()
    del vals, b, props, coords
    torch.cuda.empty_cache()
