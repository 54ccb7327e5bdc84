"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same
inputs.  Tolerance: north_star's 1e-12 relative Frobenius norm on K (and on b); the kernels actually land
near 1e-15, asserted at 1e-13 so a regression shows.  Integer outputs (pattern, scatter map, compressed
pattern) are bit-exact."""
import ctypes as C

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
    from petsc_amd import slab, synth
    assert torch.cuda.is_available()
    F.load()
    return dict(torch=torch, F=F, O=O, slab=slab, synth=synth, dev=torch.device("cuda:0"))


def make_case(env, kind, dims, perturb=True, vary_material=False):
    """Inputs + oracle system for one problem kind."""
    O, F = env["O"], env["F"]
    rng = np.random.default_rng(7)
    if kind == "p3":
        M, N, P = dims
        coords = O.uniform_coords(3, M, N, P)
        if perturb:
            coords = O.perturb_coords(coords, 3, M, N, P)
        rp, ci = O.pattern(3, 1, M, N, P)
        K, b = O.assemble_poisson3d(M, N, P, coords, rp, ci)
        return dict(problem=F.POISSON3D_Q1, dims=(M, N, P), coords=coords, props=None, stride=0,
                    bc=F.BC_POISSON_INLOOP, rp=rp, ci=ci, K=K, b=b)
    if kind == "e2":
        M, N = dims
        coords = O.uniform_coords(2, M, N)
        if perturb:
            coords = O.perturb_coords(coords, 2, M, N)
        rp, ci = O.pattern(2, 2, M, N)
        props = O.props_e2(M - 1, N - 1, coords).reshape(-1, 24)
        props[:, 16:24] = rng.uniform(-1, 1, size=(props.shape[0], 8))  # body force: exercise the RHS
        if vary_material:
            props[:, 8:12] = rng.uniform(0.5, 2.0, size=(props.shape[0], 4))
            props[:, 12:16] = rng.uniform(0.1, 0.4, size=(props.shape[0], 4))
        props = props.reshape(-1)
        K, b = O.assemble_elastic2d(M, N, coords, props, rp, ci)
        O.bc_e2(M, N, rp, ci, K, b)
        return dict(problem=F.PLANE_STRAIN2D_Q1, dims=(M, N, 1), coords=coords, props=props, stride=24,
                    bc=F.BC_E2_COMPRESSION, rp=rp, ci=ci, K=K, b=b)
    M, N, P = dims
    coords = O.uniform_coords(3, M, N, P)
    if perturb:
        coords = O.perturb_coords(coords, 3, M, N, P)
    rp, ci = O.pattern(3, 3, M, N, P)
    props = O.props_s3(M - 1, N - 1, P - 1, coords).reshape(-1, 400)
    if vary_material:
        props[:, 96:104] = rng.uniform(50, 150, size=(props.shape[0], 8))
        props[:, 104:112] = rng.uniform(0.1, 0.4, size=(props.shape[0], 8))
        props[:, 112:136] = rng.uniform(50, 150, size=(props.shape[0], 24))
        props[:, 136:160] = rng.uniform(0.1, 0.4, size=(props.shape[0], 24))
        props[:, 184:400] = rng.uniform(-1, 1, size=(props.shape[0], 216))
    props = props.reshape(-1)
    K, b = O.assemble_elastic3d(M, N, P, coords, props, rp, ci)
    O.bc_s3(M, N, P, coords, rp, ci, K, b)
    return dict(problem=F.ELASTIC3D_Q1, dims=(M, N, P), coords=coords, props=props, stride=400,
                bc=F.BC_S3_ALLFACES_EXACT, rp=rp, ci=ci, K=K, b=b)


def gpu_assemble(env, case, host_buffers=False, bc=None, options=None):
    torch, F, dev = env["torch"], env["F"], env["dev"]
    M, N, P = case["dims"]
    to = (lambda a: a) if host_buffers else (lambda a: torch.from_numpy(a).to(dev))
    plan = F.Plan(case["problem"], M, N, P, coords=to(case["coords"]), options=options)
    if host_buffers:
        vals, b = np.full(plan.nnz, np.nan), np.full(plan.nrows, np.nan)
    else:
        vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
        b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
    mat = None if case["props"] is None else to(case["props"])
    plan.assemble_system(vals, b, material=mat, stride=case["stride"], bc=case["bc"] if bc is None else bc)
    if not host_buffers:
        vals, b = vals.cpu().numpy(), b.cpu().numpy()
    return plan, vals, b


CASES = [
    ("p3", (5, 5, 5)), ("p3", (9, 6, 7)), ("p3", (2, 2, 2)), ("p3", (3, 2, 10)), ("p3", (17, 13, 6)),
    ("e2", (2, 2)), ("e2", (9, 7)), ("e2", (40, 21)), ("e2", (17, 5)), ("e2", (3, 50)),
    ("s3", (2, 2, 2)), ("s3", (4, 4, 4)), ("s3", (8, 6, 7)), ("s3", (5, 9, 3)), ("s3", (13, 4, 6)),
    ("s3", (38, 5, 36)),
    # odd mid-size meshes: several tiles / runs / chunks in every direction, none a multiple of the tile sizes
    ("p3", (131, 37, 70)), ("e2", (301, 137)), ("s3", (45, 23, 36)),
]


@pytest.mark.parametrize("kind,dims", CASES)
@pytest.mark.parametrize("perturb", [False, True])
def test_system_matches_oracle(env, kind, dims, perturb):
    case = make_case(env, kind, dims, perturb=perturb)
    plan, vals, b = gpu_assemble(env, case)
    rp, ci = plan.pattern()
    assert np.array_equal(rp, case["rp"]) and np.array_equal(ci, case["ci"])  # bit-exact pattern (A12)
    assert not np.isnan(vals).any() and not np.isnan(b).any()                  # every slot written
    assert rel(vals, case["K"]) <= TOL
    assert rel(b, case["b"]) <= TOL
    plan.close()


@pytest.mark.parametrize("kind,dims", [("e2", (12, 9)), ("s3", (6, 5, 7))])
def test_per_gauss_point_material_and_loads(env, kind, dims):
    case = make_case(env, kind, dims, vary_material=True)
    plan, vals, b = gpu_assemble(env, case)
    assert rel(vals, case["K"]) <= TOL and rel(b, case["b"]) <= TOL
    plan.close()


@pytest.mark.parametrize("kind,dims", [("p3", (6, 5, 4)), ("e2", (10, 6)), ("s3", (5, 4, 4))])
def test_host_pointers_are_staged(env, kind, dims):
    """The drivers hand in MatSeqAIJGetArray / VecGetArray host pointers."""
    case = make_case(env, kind, dims)
    plan, vals, b = gpu_assemble(env, case, host_buffers=True)
    assert rel(vals, case["K"]) <= TOL and rel(b, case["b"]) <= TOL
    plan.close()


def test_uniform_material_equals_constant_aos(env):
    torch, F, dev, O = env["torch"], env["F"], env["dev"], env["O"]
    case = make_case(env, "s3", (6, 6, 5))
    M, N, P = case["dims"]
    plan = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=torch.from_numpy(case["coords"]).to(dev))
    vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
    plan.assemble_system(vals, None, material=None, E=90.0, nu=0.28, bc=F.BC_S3_ALLFACES_EXACT)
    assert rel(vals.cpu().numpy(), case["K"]) <= TOL
    plan.close()


def test_no_bc_gives_the_raw_operator(env):
    O, F = env["O"], env["F"]
    M, N, P = 5, 4, 6
    coords = O.perturb_coords(O.uniform_coords(3, M, N, P), 3, M, N, P)
    props = O.props_s3(M - 1, N - 1, P - 1, coords)
    rp, ci = O.pattern(3, 3, M, N, P)
    K, b = O.assemble_elastic3d(M, N, P, coords, props, rp, ci)
    case = dict(problem=F.ELASTIC3D_Q1, dims=(M, N, P), coords=coords, props=props, stride=400, bc=F.BC_NONE)
    plan, vals, bb = gpu_assemble(env, case)
    assert rel(vals, K) <= TOL and rel(bb, b) <= TOL
    plan.close()


@pytest.mark.parametrize("kind,dims", [("p3", (4, 3, 5)), ("e2", (6, 4)), ("s3", (3, 4, 3))])
def test_scatter_map_is_the_reference_index_map(env, kind, dims):
    """A13: position of every (element, local row, local col) = where MatSetValuesStencil would add,
    found here by searching the sorted CSR row."""
    case = make_case(env, kind, dims, perturb=False)
    plan, _, _ = gpu_assemble(env, case)
    smap = plan.scatter_map()
    M, N, P = case["dims"]
    nsd, dof = plan.nsd, plan.dof
    rp, ci = case["rp"], case["ci"]
    hd = [(0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1), (1, 0, 1)]
    qd = [(0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 0)]
    offs = hd if nsd == 3 else qd
    e = 0
    for ek in range(max(P - 1, 1)):
        for ej in range(N - 1):
            for ei in range(M - 1):
                eq = [dof * ((ei + o[0]) + M * ((ej + o[1]) + N * (ek + o[2]))) + c for o in offs for c in range(dof)]
                for r, row in enumerate(eq):
                    cols = ci[rp[row]:rp[row + 1]]
                    want = rp[row] + np.searchsorted(cols, eq)
                    assert np.array_equal(smap[e, r], want)
                e += 1
    assert e == plan.nel
    plan.close()


@pytest.mark.parametrize("kind,dims", [("e2", (7, 5)), ("s3", (4, 3, 4))])
def test_compressed_bc_rows_match_petsc_matzerorows_default(env, kind, dims):
    O = env["O"]
    case = make_case(env, kind, dims)
    plan, vals, _ = gpu_assemble(env, case)
    M, N, P = case["dims"]
    flags = O.bc_flags("e2" if kind == "e2" else "s3", M, N, P)
    rp2, ci2, v2 = O.compress_zeroed_rows(flags, case["rp"], case["ci"], case["K"])
    grp, gci, gv = plan.compress_bc_rows(vals, case["bc"])
    assert np.array_equal(grp, rp2) and np.array_equal(gci, ci2)
    assert rel(gv, v2) <= TOL
    plan.close()


# the last three: shapes of the S3 RHS runs kernel -- rows that are not a multiple of 16 elements (short last run,
# node column on a run boundary), more than one chunk of 32 element layers (node layers whose halves come from two
# chunks), slabs whose first element layer is owed to the rank below
@pytest.mark.parametrize("kind,dims,nslab", [("p3", (5, 4, 9), 3), ("e2", (9, 11), 2), ("s3", (4, 5, 8), 2),
                                             ("s3", (3, 3, 7), 4), ("s3", (38, 4, 36), 2), ("s3", (18, 3, 70), 3),
                                             ("s3", (50, 3, 5), 2)])
@pytest.mark.parametrize("split,perturb", [(False, True), (True, True), (True, False)])
def test_slabs_plus_interface_exchange_equal_the_global_system(env, kind, dims, nslab, split, perturb):
    """Multi-GPU path on one device: one plan per slab, the send buffer of slab r+1 handed to slab r
    (what RCCL does in bench.py), then femasm_interface_apply.  split: the send buffer is produced by
    femasm_assemble_interface ahead of the full call (the overlap path of bench.py --gpus N)."""
    torch, F, dev, slab = env["torch"], env["F"], env["dev"], env["slab"]
    # perturb=False: rectilinear mesh, i.e. the level-2 kernels and the separable P3 right-hand side on slabs
    case = make_case(env, kind, dims, perturb=perturb, vary_material=(kind != "p3" and perturb))
    M, N, P = case["dims"]
    nsd = 2 if kind == "e2" else 3
    SL = P if nsd == 3 else N
    plane_n = (M * N if nsd == 3 else M)
    plane_e = ((M - 1) * (N - 1) if nsd == 3 else (M - 1))
    ranges = slab.slab_ranges(SL, nslab)
    plans, outs, bufs = [], [], []
    for (ks, ke) in ranges:
        cks, cke = max(ks - 1, 0), min(ke + 1, SL)
        els, ele = slab.element_range(ks, ke)
        co = torch.from_numpy(case["coords"].reshape(SL, -1)[cks:cke].reshape(-1).copy()).to(dev)
        plan = F.Plan(case["problem"], M, N, P, coords=co, ks=ks, ke=ke, cks=cks, cke=cke)
        mat = None
        if case["props"] is not None:
            mat = torch.from_numpy(case["props"].reshape(-1, plane_e * case["stride"])[els:ele].reshape(-1).copy()).to(dev)
        vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
        b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
        (sp, sc), (rp_, rc) = plan.interface_buffers()
        assert (sc, rc) == slab.interface_counts(nsd, plan.dof, M, N, P, ks, ke)
        send = torch.full((max(sc, 1),), float("nan"), dtype=torch.float64, device=dev)
        recv = torch.full((max(rc, 1),), float("nan"), dtype=torch.float64, device=dev)
        plan.interface_set_buffers(send if sc else None, recv if rc else None)
        if split:
            plan.assemble_interface(material=mat, stride=case["stride"], bc=case["bc"], e_ks=els)
            sent = send.clone()  # what the network would have taken at this point
        plan.assemble_system(vals, b, material=mat, stride=case["stride"], bc=case["bc"], e_ks=els)
        if split and sc:
            assert torch.equal(sent[:sc], send[:sc])  # the full call left the buffer alone
        plans.append(plan); outs.append((vals, b)); bufs.append((send, recv, sc, rc))
    for r in range(nslab - 1):
        assert bufs[r][3] == bufs[r + 1][2]
        bufs[r][1][:bufs[r][3]].copy_(bufs[r + 1][0][:bufs[r + 1][2]])  # "RCCL": slab r+1 -> slab r
    for r in range(nslab):
        plans[r].interface_apply(outs[r][0], outs[r][1], bc=case["bc"])
    K = np.concatenate([o[0].cpu().numpy() for o in outs])
    b = np.concatenate([o[1].cpu().numpy() for o in outs])
    assert not np.isnan(K).any() and not np.isnan(b).any()
    assert rel(K, case["K"]) <= TOL and rel(b, case["b"]) <= TOL
    # per-slab pattern = global pattern rows, local rowptr, global column ids
    off = 0
    for r, (ks, ke) in enumerate(ranges):
        rp, ci = plans[r].pattern()
        lo, hi = plane_n * plans[r].dof * ks, plane_n * plans[r].dof * ke
        assert np.array_equal(rp, case["rp"][lo:hi + 1] - case["rp"][lo])
        assert np.array_equal(ci, case["ci"][case["rp"][lo]:case["rp"][hi]])
        plans[r].close()


def test_host_material_then_device_material_on_one_plan(env):
    """A staged copy of a HOST material array must not outlive its call: femasm_assemble_system with host
    material, then femasm_assemble_interface + femasm_assemble_system with a DIFFERENT device material on the same
    plan (upper slab of two), against the oracle system of the second material."""
    torch, F, dev, O, slab = env["torch"], env["F"], env["dev"], env["O"], env["slab"]
    M, N, P = 7, 6, 9
    coords = O.perturb_coords(O.uniform_coords(3, M, N, P), 3, M, N, P)
    rp, ci = O.pattern(3, 3, M, N, P)
    plane_e = (M - 1) * (N - 1)
    propsA = O.props_s3(M - 1, N - 1, P - 1, coords, E=90.0, nu=0.28)
    propsB = O.props_s3(M - 1, N - 1, P - 1, coords, E=33.0, nu=0.17).reshape(-1, 400)
    propsB[:, 184:400] *= 1.7
    propsB = propsB.reshape(-1)
    KB, bB = O.assemble_elastic3d(M, N, P, coords, propsB, rp, ci)
    O.bc_s3(M, N, P, coords, rp, ci, KB, bB)
    (ks0, ke0), (ks, ke) = slab.slab_ranges(P, 2)
    cks, cke = ks - 1, P
    els, ele = slab.element_range(ks, ke)
    co = torch.from_numpy(coords.reshape(P, -1)[cks:cke].reshape(-1).copy()).to(dev)
    plan = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=co, ks=ks, ke=ke, cks=cks, cke=cke)
    vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
    b = torch.empty(plan.nrows, dtype=torch.float64, device=dev)
    (_, sc), _ = plan.interface_buffers()
    send = torch.empty(sc, dtype=torch.float64, device=dev)
    plan.interface_set_buffers(send, None)
    hostA = propsA.reshape(-1, plane_e * 400)[els:ele].reshape(-1).copy()             # HOST array: staged
    plan.assemble_system(vals, b, material=hostA, stride=400, bc=F.BC_S3_ALLFACES_EXACT, e_ks=els)
    devB = torch.from_numpy(propsB.reshape(-1, plane_e * 400)[els:ele].reshape(-1).copy()).to(dev)
    plan.assemble_interface(material=devB, stride=400, bc=F.BC_S3_ALLFACES_EXACT, e_ks=els)
    plan.assemble_system(vals, b, material=devB, stride=400, bc=F.BC_S3_ALLFACES_EXACT, e_ks=els)
    # lower slab with material B, fed with the upper slab's send buffer
    co0 = torch.from_numpy(coords.reshape(P, -1)[0:ke0 + 1].reshape(-1).copy()).to(dev)
    plan0 = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=co0, ks=ks0, ke=ke0, cks=0, cke=ke0 + 1)
    e0, e1 = slab.element_range(ks0, ke0)
    dev0 = torch.from_numpy(propsB.reshape(-1, plane_e * 400)[e0:e1].reshape(-1).copy()).to(dev)
    v0 = torch.empty(plan0.nnz, dtype=torch.float64, device=dev)
    b0 = torch.empty(plan0.nrows, dtype=torch.float64, device=dev)
    (_, _), (_, rc) = plan0.interface_buffers()
    assert rc == sc
    recv = send.clone()                                                               # "RCCL"
    plan0.interface_set_buffers(None, recv)
    plan0.assemble_system(v0, b0, material=dev0, stride=400, bc=F.BC_S3_ALLFACES_EXACT, e_ks=e0)
    plan0.interface_apply(v0, b0, bc=F.BC_S3_ALLFACES_EXACT)
    K = np.concatenate([v0.cpu().numpy(), vals.cpu().numpy()])
    bb = np.concatenate([b0.cpu().numpy(), b.cpu().numpy()])
    assert rel(K, KB) <= TOL and rel(bb, bB) <= TOL
    plan.close(); plan0.close()


@pytest.mark.parametrize("kind,dims", [("p3", (20, 9, 11)), ("e2", (70, 19)), ("s3", (21, 7, 18))])
@pytest.mark.parametrize("perturb", [False, True])
def test_asynchronous_assembly_is_final_in_stream_order(env, kind, dims, perturb):
    """femasm_plan_set_async: no host wait inside femasm_assemble_system; on a mesh that breaks the fast kernels'
    premise the general kernels queued behind them must have produced the result by the time the stream gets there,
    and the plan's level drops when the flags are read (femasm_plan_sync)."""
    torch, F, dev = env["torch"], env["F"], env["dev"]
    case = make_case(env, kind, dims, perturb=perturb)
    M, N, P = case["dims"]
    plan = F.Plan(case["problem"], M, N, P, coords=torch.from_numpy(case["coords"]).to(dev))
    plan.set_async(True)
    vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
    b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
    mat = None if case["props"] is None else torch.from_numpy(case["props"]).to(dev)
    for _ in range(2):   # the second call starts by reading the first one's flags
        plan.assemble_system(vals, b, material=mat, stride=case["stride"], bc=case["bc"])
    plan.sync()
    # (a broken box premise leaves the elastic kind at level 1: any hexahedra with one (E, nu) per element)
    assert plan.fast_path() == ((1 if kind == "s3" else 0) if perturb else 2)
    assert rel(vals.cpu().numpy(), case["K"]) <= TOL and rel(b.cpu().numpy(), case["b"]) <= TOL
    plan.close()


def test_asynchronous_assembly_reports_geometry_errors_at_sync(env):
    torch, F, dev, O = env["torch"], env["F"], env["dev"], env["O"]
    M = 5
    coords = O.uniform_coords(3, M, M, M).reshape(-1, 3)
    coords[2 + M * (2 + M * 2)] = coords[3 + M * (3 + M * 3)] + 0.3   # a node far outside its cell: inverted elements
    plan = F.Plan(F.POISSON3D_Q1, M, M, M, coords=torch.from_numpy(coords.reshape(-1)).to(dev))
    plan.set_async(True)
    vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
    plan.assemble_system(vals, None, bc=F.BC_POISSON_INLOOP)      # returns at once, no error yet
    with pytest.raises(F.FemasmError) as ei:
        plan.sync()
    assert ei.value.code == 3
    plan.close()


@pytest.mark.parametrize("kind,dims", [("p3", (6, 5, 4)), ("e2", (9, 7)), ("s3", (5, 4, 4))])
@pytest.mark.parametrize("nslab", [1, 2])
def test_opt_in_symmetric_elimination(env, kind, dims, nslab):
    """femasm_bc.symmetric / femasm_eliminate_columns (NOT reference parity: the reference zeroes rows only): against
    the elimination done in numpy on the oracle's rows-only system, and against scipy: same solution, symmetric K."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    torch, F, dev, slab = env["torch"], env["F"], env["dev"], env["slab"]
    case = make_case(env, kind, dims)
    M, N, P = case["dims"]
    rp, ci = case["rp"], case["ci"]
    n = len(rp) - 1
    A = sp.csr_matrix((case["K"], ci, rp), shape=(n, n)).toarray()
    b = case["b"].copy()
    isbc = np.array([A[r, r] != 0 and np.count_nonzero(A[r]) == 1 and (kind == "p3" or A[r, r] == 1.0) for r in range(n)])
    if kind == "p3":   # Poisson boundary rows: valence * identity, value 0
        g = np.zeros(n)
    else:
        g = np.where(isbc, b, 0.0)
    As, bs = A.copy(), b.copy()
    for c in np.nonzero(isbc)[0]:
        rows = ~isbc
        bs[rows] -= A[rows, c] * g[c]
        As[rows, c] = 0.0
    nsd = 2 if kind == "e2" else 3
    SL = P if nsd == 3 else N
    if nslab == 1:
        plan = F.Plan(case["problem"], M, N, P, coords=torch.from_numpy(case["coords"]).to(dev))
        vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
        bb = torch.empty(plan.nrows, dtype=torch.float64, device=dev)
        mat = None if case["props"] is None else torch.from_numpy(case["props"]).to(dev)
        plan.assemble_system(vals, bb, material=mat, stride=case["stride"], bc=case["bc"], symmetric=True)
        plan.close()
        Kg, bg = vals.cpu().numpy(), bb.cpu().numpy()
    else:
        plane_e = ((M - 1) * (N - 1) if nsd == 3 else (M - 1))
        outs, bufs, plans = [], [], []
        for (ks, ke) in slab.slab_ranges(SL, 2):
            cks, cke = max(ks - 1, 0), min(ke + 1, SL)
            els, ele = slab.element_range(ks, ke)
            co = torch.from_numpy(case["coords"].reshape(SL, -1)[cks:cke].reshape(-1).copy()).to(dev)
            plan = F.Plan(case["problem"], M, N, P, coords=co, ks=ks, ke=ke, cks=cks, cke=cke)
            mat = None
            if case["props"] is not None:
                mat = torch.from_numpy(case["props"].reshape(-1, plane_e * case["stride"])[els:ele].reshape(-1).copy()).to(dev)
            vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
            bb = torch.empty(plan.nrows, dtype=torch.float64, device=dev)
            (_, sc), (_, rc) = plan.interface_buffers()
            send = torch.empty(max(sc, 1), dtype=torch.float64, device=dev)
            recv = torch.empty(max(rc, 1), dtype=torch.float64, device=dev)
            plan.interface_set_buffers(send if sc else None, recv if rc else None)
            with pytest.raises(F.FemasmError):   # a slab's rows are not complete inside assemble_system
                plan.assemble_system(vals, bb, material=mat, stride=case["stride"], bc=case["bc"], e_ks=els, symmetric=True)
            plan.assemble_system(vals, bb, material=mat, stride=case["stride"], bc=case["bc"], e_ks=els)
            plans.append(plan); outs.append((vals, bb)); bufs.append((send, recv, sc, rc))
        bufs[0][1][:bufs[0][3]].copy_(bufs[1][0][:bufs[1][2]])
        for r in range(2):
            plans[r].interface_apply(outs[r][0], outs[r][1], bc=case["bc"])
            plans[r].eliminate_columns(outs[r][0], outs[r][1], case["bc"])
            plans[r].close()
        Kg = np.concatenate([o[0].cpu().numpy() for o in outs])
        bg = np.concatenate([o[1].cpu().numpy() for o in outs])
    Ag = sp.csr_matrix((Kg, ci, rp), shape=(n, n)).toarray()
    assert np.linalg.norm(Ag - As) <= TOL * np.linalg.norm(As) and np.linalg.norm(bg - bs) <= TOL * max(np.linalg.norm(bs), 1.0)
    assert np.abs(Ag - Ag.T).max() <= 1e-12 * np.abs(Ag).max()                      # symmetric now
    x_rows = spla.spsolve(sp.csc_matrix(A), b)                                        # the reference's system
    x_sym = spla.spsolve(sp.csc_matrix(Ag), bg)
    assert np.linalg.norm(x_rows - x_sym) <= 1e-9 * max(np.linalg.norm(x_rows), 1e-30)


@pytest.mark.parametrize("dims,perturb", [((7, 6, 5), False), ((7, 6, 5), True), ((20, 9, 35), False), ((4, 3, 9), True)])
@pytest.mark.parametrize("nslab", [1, 2])
def test_manufactured_loads_layout_equals_the_records_the_drivers_fill(env, dims, perturb, nslab):
    """FEMASM_MAT_MANUFACTURED: f and grad u evaluated in the right-hand-side kernel; b (and K) against the oracle
    assembling from the oracle's own fill of the records (S3:1243-1363), 1e-13."""
    torch, F, dev, slab = env["torch"], env["F"], env["dev"], env["slab"]
    case = make_case(env, "s3", dims, perturb=perturb)
    M, N, P = case["dims"]
    Ks, bs = [], []
    for (ks, ke) in slab.slab_ranges(P, nslab):
        cks, cke = max(ks - 1, 0), min(ke + 1, P)
        co = torch.from_numpy(case["coords"].reshape(P, -1)[cks:cke].reshape(-1).copy()).to(dev)
        plan = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=co, ks=ks, ke=ke, cks=cks, cke=cke)
        vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
        b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
        plan.assemble_system(vals, b, E=90.0, nu=0.28, bc=F.BC_S3_ALLFACES_EXACT, manufactured=True)
        Ks.append((plan, vals, b))
    for r in range(nslab - 1):
        (_, sc), _ = Ks[r + 1][0].interface_buffers()
        recv = torch.empty(sc, dtype=torch.float64, device=dev)
        send = torch.empty(sc, dtype=torch.float64, device=dev)   # caller-owned wire buffer: assemble again into it
        Ks[r + 1][0].interface_set_buffers(send, None)
        Ks[r + 1][0].assemble_system(Ks[r + 1][1], Ks[r + 1][2], E=90.0, nu=0.28, bc=F.BC_S3_ALLFACES_EXACT, manufactured=True)
        recv.copy_(send)
        Ks[r][0].interface_set_buffers(None, recv)
    for r in range(nslab):
        Ks[r][0].interface_apply(Ks[r][1], Ks[r][2], bc=F.BC_S3_ALLFACES_EXACT)
    K = np.concatenate([k[1].cpu().numpy() for k in Ks])
    b = np.concatenate([k[2].cpu().numpy() for k in Ks])
    for k in Ks:
        k[0].close()
    assert rel(K, case["K"]) <= TOL and rel(b, case["b"]) <= TOL, (rel(K, case["K"]), rel(b, case["b"]))


def test_compress_bc_rows_rejects_the_poisson_kind(env):
    """P3 never calls MatZeroRows: its boundary diagonal is the element valence, which the compressed form (diag 1)
    would misstate."""
    F = env["F"]
    case = make_case(env, "p3", (4, 4, 4))
    plan, vals, _ = gpu_assemble(env, case)
    with pytest.raises(F.FemasmError):
        plan.compress_bc_rows(vals, F.BC_POISSON_INLOOP)
    plan.close()


def test_results_are_bitwise_reproducible(env):
    case = make_case(env, "s3", (9, 8, 7))
    p1, v1, b1 = gpu_assemble(env, case)
    p2, v2, b2 = gpu_assemble(env, case)
    assert np.array_equal(v1, v2) and np.array_equal(b1, b2)
    p1.close(); p2.close()


@pytest.mark.parametrize("perturb", [False, True])
def test_s3_rhs_persistent_waves_walk_several_tasks(env, monkeypatch, perturb):
    """k_rhs_elem_elastic_runs is launched as one resident set of waves that stride over the (chunk, row, run) tasks;
    small meshes give every wave one task, so the grid is capped here (option "rhs_grid") to make each wave walk many,
    on the box path and on the general one"""
    torch, F, dev, O = env["torch"], env["F"], env["dev"], env["O"]
    M, N, P = 35, 9, 40          # 3 runs x 8 rows x 2 chunks = 48 tasks
    coords = O.uniform_coords(3, M, N, P)
    if perturb:
        coords = O.perturb_coords(coords, 3, M, N, P)
    props = O.props_s3(M - 1, N - 1, P - 1, coords)
    rp, ci = O.pattern(3, 3, M, N, P)
    K, b = O.assemble_elastic3d(M, N, P, coords, props, rp, ci)
    O.bc_s3(M, N, P, coords, rp, ci, K, b)
    plan = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=torch.from_numpy(coords).to(dev), options={"rhs_grid": 5})
    gb = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
    plan.assemble_system(None, gb, material=torch.from_numpy(props).to(dev), stride=400, bc=F.BC_S3_ALLFACES_EXACT)
    plan.close()
    assert rel(gb.cpu().numpy(), b) <= TOL


@pytest.mark.parametrize("variant", ["units=1", "units=3", "units=64", "rowsweep"])
@pytest.mark.parametrize("dims", [(45, 23, 36), (38, 5, 7), (10, 4, 4)])
def test_s3_box_kernels_unit_and_rowsweep(env, monkeypatch, variant, dims):
    """Elastic K on a rectilinear mesh (level 2) has two kernels: k_assemble_s3_unit (default; a wave walks
    "s3_units" x-consecutive 8-node units, chosen from the mesh size otherwise, so small meshes only ever see 1)
    and the persistent k_assemble_rowsweep (option "s3_rowsweep").  Both against the oracle, with a material that
    differs from element to element, on meshes whose x range is not a multiple of the unit."""
    torch, F = env["torch"], env["F"]
    case = make_case(env, "s3", dims, perturb=False)
    rng = np.random.default_rng(5)
    props = case["props"].reshape(-1, 400).copy()
    props[:, 96:104] = rng.uniform(50, 150, size=(props.shape[0], 1))     # one (E, nu) per element
    props[:, 104:112] = rng.uniform(0.1, 0.4, size=(props.shape[0], 1))
    case["props"] = props.reshape(-1)
    M, N, P = dims
    O = env["O"]
    K, b = O.assemble_elastic3d(M, N, P, case["coords"], case["props"], case["rp"], case["ci"])
    O.bc_s3(M, N, P, case["coords"], case["rp"], case["ci"], K, b)
    opts = {"s3_rowsweep": 1} if variant == "rowsweep" else {"s3_units": int(variant.split("=")[1])}
    plan, vals, gb = gpu_assemble(env, case, options=opts)
    assert plan.fast_path() == 2
    assert not np.isnan(vals).any()
    assert rel(vals, K) <= TOL and rel(gb, b) <= TOL
    plan.close()


@pytest.mark.parametrize("kind,dims", [("s3", (21, 7, 18)), ("s3", (6, 5, 4)), ("e2", (70, 19)), ("e2", (5, 4))])
@pytest.mark.parametrize("mesh", ["uniform", "perturbed"])
@pytest.mark.parametrize("variant", ["default", "rowsweep"])
def test_element_soa_material_gives_the_same_k_as_the_records(env, monkeypatch, kind, dims, mesh, variant):
    """FEMASM_MAT_ELEMENT_SOA (A14's compact form: E per element, then nu per element) against the AoS records that hold
    the same pair at every Gauss point: the same K bit for bit, on the box kernels (both S3 ones), the general kernels and
    the boundary rows; b holds the Dirichlet data only, as with a uniform material.  Also y = K x matrix-free."""
    torch, F, dev = env["torch"], env["F"], env["dev"]
    if variant == "rowsweep":
        if kind != "s3" or mesh != "uniform":
            pytest.skip("the row-sweep variant only differs for S3 boxes")
    opts = {"s3_rowsweep": 1} if variant == "rowsweep" else None
    case = make_case(env, kind, dims, perturb=(mesh == "perturbed"))
    rng = np.random.default_rng(11)
    st = case["stride"]
    props = case["props"].reshape(-1, st).copy()
    nel = props.shape[0]
    Ee, nue = rng.uniform(50, 150, size=nel), rng.uniform(0.1, 0.4, size=nel)
    oE, oN, ng = (96, 104, 8) if kind == "s3" else (8, 12, 4)
    props[:, oE:oE + ng] = Ee[:, None]
    props[:, oN:oN + ng] = nue[:, None]
    dprops = torch.from_numpy(props.reshape(-1)).to(dev)
    pad = 5                                            # plane distance larger than the element count
    planes = torch.from_numpy(np.concatenate([Ee, np.zeros(pad), nue])).to(dev)
    coords = torch.from_numpy(case["coords"]).to(dev)
    plan = F.Plan(case["problem"], *case["dims"], coords=coords, options=opts)
    v_aos = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
    v_soa = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
    b_soa = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
    b_uni = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
    plan.assemble_system(v_aos, None, material=dprops, stride=st, bc=case["bc"])
    lvl = plan.fast_path()
    plan.assemble_system(v_soa, b_soa, material=planes, stride=nel + pad, bc=case["bc"], element_soa=True)
    assert plan.fast_path() == lvl and lvl == (2 if mesh == "uniform" else (1 if kind == "s3" else 0))
    assert not torch.isnan(v_soa).any()
    assert torch.equal(v_aos, v_soa)
    plan.assemble_system(None, b_uni, material=None, E=1.0, nu=0.3, bc=case["bc"])
    assert torch.equal(b_soa, b_uni)
    # matrix-free product
    x = torch.from_numpy(rng.uniform(-1, 1, size=plan.nrows)).to(dev)
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    plan.apply_operator(x, y1, material=dprops, stride=st, bc=case["bc"])
    plan.apply_operator(x, y2, material=planes, stride=nel + pad, bc=case["bc"], element_soa=True)
    assert torch.equal(y1, y2)
    # too small a plane distance, host planes: refused
    with pytest.raises(F.FemasmError):
        plan.assemble_system(v_soa, None, material=planes, stride=nel - 1, bc=case["bc"], element_soa=True)
    with pytest.raises(F.FemasmError):
        plan.assemble_system(v_soa, None, material=planes.cpu().numpy(), stride=nel + pad, bc=case["bc"], element_soa=True)
    plan.close()


@pytest.mark.parametrize("kernel", ["ke", "brick"])
@pytest.mark.parametrize("dims", [(70, 19), (17, 9), (4, 3), (3, 40), (131, 37)])
def test_e2_general_kernels_element_matrix_and_brick(env, monkeypatch, kernel, dims):
    """Plane strain on general quadrilaterals has two kernels for the interior nodes: k_assemble_ke_e2 (default: the
    ten unique 2 x 2 blocks of each element matrix formed once, node rows gathered from LDS; 15 x 7 node tiles) and the
    brick kernel k_assemble_rows (option "e2_brick").  Both against the oracle, per-Gauss-point material, on perturbed
    meshes whose extents are not multiples of the tile."""
    case = make_case(env, "e2", dims, perturb=True, vary_material=True)
    plan, vals, gb = gpu_assemble(env, case, options={"e2_brick": 1} if kernel == "brick" else None)
    assert plan.fast_path() == 0
    assert not np.isnan(vals).any()
    assert rel(vals, case["K"]) <= TOL and rel(gb, case["b"]) <= TOL
    plan.close()


@pytest.mark.parametrize("mesh", ["uniform", "perturbed"])
def test_element_soa_material_on_an_upper_slab(env, mesh):
    """FEMASM_MAT_ELEMENT_SOA on the upper of two slabs (planes that start at the slab's first element layer, e_ks):
    the slab's rows and the rows it owes to the rank below (send buffer) are the bits the AoS records give."""
    torch, F, dev, O, slab = env["torch"], env["F"], env["dev"], env["O"], env["slab"]
    M, N, P = 19, 6, 11
    coords = O.uniform_coords(3, M, N, P)
    if mesh == "perturbed":
        coords = O.perturb_coords(coords, 3, M, N, P)
    plane_e = (M - 1) * (N - 1)
    rng = np.random.default_rng(3)
    nel = plane_e * (P - 1)
    Ee, nue = rng.uniform(50, 150, size=nel), rng.uniform(0.1, 0.4, size=nel)
    props = O.props_s3(M - 1, N - 1, P - 1, coords).reshape(-1, 400)
    props[:, 96:104] = Ee[:, None]
    props[:, 104:112] = nue[:, None]
    (_, _), (ks, ke) = slab.slab_ranges(P, 2)
    cks, cke = ks - 1, P
    els, ele = slab.element_range(ks, ke)
    co = torch.from_numpy(coords.reshape(P, -1)[cks:cke].reshape(-1).copy()).to(dev)
    plan = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=co, ks=ks, ke=ke, cks=cks, cke=cke)
    (_, sc), _ = plan.interface_buffers()
    send = torch.zeros(sc, dtype=torch.float64, device=dev)
    plan.interface_set_buffers(send, None)
    aos = torch.from_numpy(props[els * plane_e:ele * plane_e].reshape(-1).copy()).to(dev)
    held = (ele - els) * plane_e
    planes = torch.from_numpy(np.concatenate([Ee[els * plane_e:ele * plane_e], nue[els * plane_e:ele * plane_e]])).to(dev)
    out = []
    # (one call first: the plan settles on the kernel level this mesh allows, so that both layouts below run the same
    # kernels -- levels differ in rounding, layouts must not differ at all)
    plan.assemble_system(torch.empty(plan.nnz, dtype=torch.float64, device=dev), None, material=aos, stride=400,
                         bc=F.BC_S3_ALLFACES_EXACT, e_ks=els)
    for mat, kw in ((aos, dict(stride=400)), (planes, dict(stride=held, element_soa=True))):
        v = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
        send.zero_()
        plan.assemble_interface(material=mat, bc=F.BC_S3_ALLFACES_EXACT, e_ks=els, **kw)
        plan.assemble_system(v, None, material=mat, bc=F.BC_S3_ALLFACES_EXACT, e_ks=els, **kw)
        plan.sync()
        out.append((v.clone(), send.clone()))
    assert not torch.isnan(out[1][0]).any()
    nk = sc - 3 * M * N     # the send buffer ends with the layer's b slice, which only the records can fill (loads)
    assert torch.equal(out[0][0], out[1][0])
    assert torch.equal(out[0][1][:nk], out[1][1][:nk])
    assert out[0][1][:nk].abs().max() > 0
    plan.close()


@pytest.mark.parametrize("kernel", ["hex", "hex2w", "ke", "sweep"])
@pytest.mark.parametrize("dims", [(33, 9, 21), (17, 5, 4), (4, 3, 5), (47, 12, 35), (18, 10, 3), (3, 3, 40)])
def test_p3_general_kernels_tensor_product_element_matrix_and_node_row(env, monkeypatch, kernel, dims):
    """Poisson K on general hexahedra has four kernels: k_assemble_p3_hex4 (default since round 3: the element matrix in
    tensor-product form, 15 x 7 node tiles, element waves and row waves, two tiles per workgroup), k_assemble_p3_hex
    (option "p3_hex2w": the same arithmetic, every wave does every phase), k_assemble_ke_p3 (option "p3_sweep" = 2: the
    36 unique entries of each element matrix from literal reference gradients, 15 x 3 node tiles) and the node-row
    kernel k_assemble_sweep_p3 ("p3_sweep" = 1).  All against the oracle on perturbed meshes whose extents are not
    multiples of any tile (odd and even numbers of tiles in x: the second tile of the last workgroup empty or not)."""
    case = make_case(env, "p3", dims, perturb=True)
    opts = {"hex": None, "hex2w": {"p3_hex2w": 1}, "ke": {"p3_sweep": 2}, "sweep": {"p3_sweep": 1}}[kernel]
    plan, vals, gb = gpu_assemble(env, case, options=opts)
    assert plan.fast_path() == 0
    assert not np.isnan(vals).any()
    assert rel(vals, case["K"]) <= TOL and rel(gb, case["b"]) <= TOL
    plan.close()


@pytest.mark.parametrize("dims", [(140, 7), (67, 12), (5, 40)])
def test_e2_one_pass_record_stream_and_lane_loads(env, dims):
    """k_assemble_e2_rect<one pass>: the records of an element row as one stream through LDS (default since round 3) and
    loaded lane by lane (option "e2_lane_loads"): the same K and b bit for bit (the arithmetic is the same), both within
    the tolerance of the oracle; runs of 64, 64 and 10 nodes, 64 and 1, and a single short run"""
    case = make_case(env, "e2", dims, perturb=False, vary_material=False)
    plan, vals, gb = gpu_assemble(env, case)
    assert plan.fast_path() == 2
    plan2, vals2, gb2 = gpu_assemble(env, case, options={"e2_lane_loads": 1})
    assert plan2.fast_path() == 2
    assert rel(vals, case["K"]) <= TOL and rel(gb, case["b"]) <= TOL
    assert np.array_equal(vals, vals2) and np.array_equal(gb, gb2)
    plan.close(); plan2.close()


def test_p3_general_load_tensor_product_form_and_as_stated(env):
    """the Poisson load on general hexahedra: the default (Hadamard geometry, one-polynomial sinpi) and the
    per-Gauss-point statement (option "rhs_simple", the library's sinpi) both match the oracle; coordinates outside
    [0, 1] (a shifted, scaled mesh: arguments of sin(pi x) up to 7) exercise the argument reduction"""
    O = env["O"]
    M, N, P = 12, 7, 9
    for shift, scale in ((0.0, 1.0), (-3.25, 2.5), (4.5, 1.0)):
        case = make_case(env, "p3", (M, N, P), perturb=True)
        co = np.ascontiguousarray(case["coords"] * scale + shift)
        K, b = O.assemble_poisson3d(M, N, P, co, case["rp"], case["ci"])
        case2 = dict(case, coords=co, K=K, b=b)
        for opts in (None, {"rhs_simple": 1}):
            plan, vals, gb = gpu_assemble(env, case2, options=opts)
            assert rel(gb, b) <= TOL, (shift, scale, opts, rel(gb, b))
            assert rel(vals, K) <= TOL
            plan.close()


@pytest.mark.parametrize("zc", [1, 2, 5])
def test_p3_hex_node_layers_per_workgroup(env, zc):
    """k_assemble_p3_hex marches zc node layers per workgroup (option "zc"; the launch picks it from the mesh): chunks of
    one, two and five layers over 11 interior layers give the same K bit for bit as the launch's own choice"""
    case = make_case(env, "p3", (20, 11, 13), perturb=True)
    plan0, vals0, _ = gpu_assemble(env, case)
    plan2, vals2w, _ = gpu_assemble(env, case, options={"p3_hex2w": 1})
    assert np.array_equal(vals2w, vals0)                                 # the two forms do the same arithmetic
    plan2.close()
    plan, vals, gb = gpu_assemble(env, case, options={"zc": zc})
    assert rel(vals, case["K"]) <= TOL and rel(gb, case["b"]) <= TOL
    assert np.array_equal(vals, vals0)
    plan.close(); plan0.close()


def test_p3_hex_inverted_element_is_reported(env):
    """det J <= 0 at a Gauss point of the Poisson kind on a distorted mesh -> FEMASM_ERR_GEOMETRY after the call"""
    torch, F, dev, O = env["torch"], env["F"], env["dev"], env["O"]
    M = N = P = 7
    coords = O.perturb_coords(O.uniform_coords(3, M, N, P), 3, M, N, P).reshape(-1, 3)
    coords[2 + M * (3 + N * 3)] = coords[4 + M * (3 + N * 3)] + np.array([0.01, 0.0, 0.0])  # pushed past its neighbour
    plan = F.Plan(F.POISSON3D_Q1, M, N, P, coords=torch.from_numpy(coords.reshape(-1).copy()).to(dev))
    vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
    b = torch.empty(plan.nrows, dtype=torch.float64, device=dev)
    with pytest.raises(F.FemasmError) as ei:
        plan.assemble_system(vals, b, bc=F.BC_POISSON_INLOOP)
    assert ei.value.code == 3 and "det(J)" in str(ei.value)
    plan.close()


@pytest.mark.parametrize("kind", ["e2", "s3"])
def test_inverted_element_is_reported_by_the_elastic_kinds(env, kind):
    """det J <= 0 at a Gauss point -> FEMASM_ERR_GEOMETRY after the call, on the general kernels the distorted mesh
    falls back to (the level-2 premise fails first) and when level 0 is set up front"""
    torch, F, dev, O = env["torch"], env["F"], env["dev"], env["O"]
    for level in (2, 0):
        if kind == "e2":
            M, N = 9, 7
            coords = O.uniform_coords(2, M, N).reshape(-1, 2)
            coords[3 + M * 3] = coords[5 + M * 3] + np.array([0.01, 0.0])   # a node pushed past its right-hand neighbour
            plan = F.Plan(F.PLANE_STRAIN2D_Q1, M, N, 1, coords=torch.from_numpy(coords.reshape(-1)).to(dev))
            props, stride, bc = O.props_e2(M - 1, N - 1, O.uniform_coords(2, M, N)), 24, F.BC_E2_COMPRESSION
        else:
            M = N = 6
            coords = O.uniform_coords(3, M, M, M).reshape(-1, 3)
            coords[2 + M * (2 + M * 2)] = coords[4 + M * (2 + M * 2)] + np.array([0.01, 0.0, 0.0])
            plan = F.Plan(F.ELASTIC3D_Q1, M, M, M, coords=torch.from_numpy(coords.reshape(-1)).to(dev))
            props, stride, bc = O.props_s3(M - 1, M - 1, M - 1, O.uniform_coords(3, M, M, M)), 400, F.BC_S3_ALLFACES_EXACT
        plan.set_fast_path(level)
        vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
        b = torch.empty(plan.nrows, dtype=torch.float64, device=dev)
        with pytest.raises(F.FemasmError) as ei:
            plan.assemble_system(vals, b, material=torch.from_numpy(props).to(dev), stride=stride, bc=bc)
        assert ei.value.code == 3 and "det(J)" in str(ei.value)
        plan.close()


def test_inverted_element_is_reported(env):
    torch, F, dev, O = env["torch"], env["F"], env["dev"], env["O"]
    M = 4
    coords = O.uniform_coords(3, M, M, M).reshape(-1, 3)
    coords[21] = coords[21] + np.array([2.0, 0.0, 0.0])  # push an interior node through its neighbours
    plan = F.Plan(F.POISSON3D_Q1, M, M, M, coords=torch.from_numpy(coords.reshape(-1)).to(dev))
    vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
    with pytest.raises(F.FemasmError) as ei:
        plan.assemble_system(vals, None, bc=F.BC_POISSON_INLOOP)
    assert ei.value.code == 3 and "det(J)" in str(ei.value)
    plan.close()


def test_mismatched_bc_kind_is_rejected(env):
    torch, F, dev, O = env["torch"], env["F"], env["dev"], env["O"]
    plan = F.Plan(F.POISSON3D_Q1, 3, 3, 3, coords=torch.from_numpy(O.uniform_coords(3, 3, 3, 3)).to(dev))
    vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
    with pytest.raises(F.FemasmError) as ei:
        plan.assemble_system(vals, None, bc=F.BC_S3_ALLFACES_EXACT)
    assert ei.value.code == 1
    plan.close()


# ---- specialised stiffness kernels (femasm_plan_set_fast_path): premise tests and fallback ----
def _s3_case_from_coords(env, dims, coords, vary_material=False):
    O, F = env["O"], env["F"]
    M, N, P = dims
    rp, ci = O.pattern(3, 3, M, N, P)
    props = O.props_s3(M - 1, N - 1, P - 1, coords).reshape(-1, 400)
    if vary_material:
        rng = np.random.default_rng(3)
        props[:, 96:104] = rng.uniform(50, 150, size=(props.shape[0], 8))
    props = props.reshape(-1)
    K, b = O.assemble_elastic3d(M, N, P, coords, props, rp, ci)
    O.bc_s3(M, N, P, coords, rp, ci, K, b)
    return dict(problem=F.ELASTIC3D_Q1, dims=(M, N, P), coords=coords, props=props, stride=400,
                bc=F.BC_S3_ALLFACES_EXACT, rp=rp, ci=ci, K=K, b=b)


def _sheared(O, M, N, P):
    """Parallelepiped elements with exactly representable coordinates (h = 1/8, dyadic shear)."""
    c = O.uniform_coords(3, M, N, P).reshape(-1, 3)
    A = np.array([[1.0, 0.25, 0.5], [0.0, 1.0, 0.25], [0.125, 0.0, 1.0]])
    return (c @ A.T).reshape(-1).copy()


def test_box_kernel_is_kept_on_uniform_anisotropic_mesh(env):
    case = make_case(env, "s3", (9, 7, 6), perturb=False)
    plan, vals, b = gpu_assemble(env, case)
    assert plan.fast_path() == 2
    assert rel(vals, case["K"]) <= TOL and rel(b, case["b"]) <= TOL
    plan.close()


def test_sheared_mesh_elastic_falls_back_poisson_stays_fast(env):
    O = env["O"]
    dims = (9, 9, 9)
    case = _s3_case_from_coords(env, dims, _sheared(O, *dims))
    plan, vals, b = gpu_assemble(env, case)
    assert plan.fast_path() == 1  # parallelepipeds are not boxes: the hexahedron kernel (one material per element)
    assert rel(vals, case["K"]) <= TOL and rel(b, case["b"]) <= TOL
    plan.close()
    # Poisson: the plan goes from level 2 straight to level 0 (since round 3 the general kernel k_assemble_p3_hex4 is
    # faster than the parallelepiped kernel of level 1); level 1, set by the caller, takes any parallelepiped and stays
    M, N, P = dims
    rp, ci = O.pattern(3, 1, M, N, P)
    K, bb = O.assemble_poisson3d(M, N, P, case["coords"], rp, ci)
    pc = dict(problem=env["F"].POISSON3D_Q1, dims=dims, coords=case["coords"], props=None, stride=0,
              bc=env["F"].BC_POISSON_INLOOP, rp=rp, ci=ci, K=K, b=bb)
    plan, vals, b = gpu_assemble(env, pc)
    assert plan.fast_path() == 0
    assert rel(vals, K) <= TOL and rel(b, bb) <= TOL
    plan.close()
    torch, F, dev = env["torch"], env["F"], env["dev"]
    plan = F.Plan(F.POISSON3D_Q1, M, N, P, coords=torch.from_numpy(case["coords"]).to(dev))
    plan.set_fast_path(1)
    v1 = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
    b1 = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
    plan.assemble_system(v1, b1, bc=F.BC_POISSON_INLOOP)
    assert plan.fast_path() == 1
    assert rel(v1.cpu().numpy(), K) <= TOL and rel(b1.cpu().numpy(), bb) <= TOL
    plan.close()


@pytest.mark.parametrize("dims", [(8, 7, 9), (14, 11, 6), (27, 10, 47), (9, 30, 5), (3, 3, 3), (45, 23, 36)])
@pytest.mark.parametrize("material", ["uniform_records", "per_element", "uniform_args"])
def test_s3_hexahedron_kernel_level_1(env, dims, material):
    """k_assemble_s3_hex2 (femasm_hex2.h): any hexahedra, one (E, nu) per element taken out of the Gauss-point loop.
    Perturbed meshes whose extents are not multiples of its 8 x 2 node tile nor of its z chunks, material the same in
    every record / different per element / passed as two numbers; the plan must stay at level 1 and K, b match the
    oracle's per-Gauss-point loops (S3:450-518)."""
    torch, F, dev, O = env["torch"], env["F"], env["dev"], env["O"]
    M, N, P = dims
    coords = O.perturb_coords(O.uniform_coords(3, M, N, P), 3, M, N, P)
    rp, ci = O.pattern(3, 3, M, N, P)
    props = O.props_s3(M - 1, N - 1, P - 1, coords).reshape(-1, 400)
    if material == "per_element":
        rng = np.random.default_rng(5)
        nel = props.shape[0]
        props[:, 96:104] = rng.uniform(50, 150, size=nel)[:, None]
        props[:, 104:112] = rng.uniform(0.1, 0.4, size=nel)[:, None]
    props = props.reshape(-1)
    K, b = O.assemble_elastic3d(M, N, P, coords, props, rp, ci)
    O.bc_s3(M, N, P, coords, rp, ci, K, b)
    plan = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=torch.from_numpy(coords).to(dev))
    plan.set_fast_path(1)
    vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
    if material == "uniform_args":
        plan.assemble_system(vals, None, material=None, E=90.0, nu=0.28, bc=F.BC_S3_ALLFACES_EXACT)
    else:
        gb = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
        plan.assemble_system(vals, gb, material=torch.from_numpy(props).to(dev), stride=400, bc=F.BC_S3_ALLFACES_EXACT)
        assert rel(gb.cpu().numpy(), b) <= TOL
    assert plan.fast_path() == 1
    v = vals.cpu().numpy()
    assert not np.isnan(v).any()
    assert rel(v, K) <= TOL
    # twice the same bits (fixed summation trees, no atomics on values)
    v2 = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
    plan.assemble_system(v2, None, material=None if material == "uniform_args" else torch.from_numpy(props).to(dev),
                         E=90.0, nu=0.28, stride=400, bc=F.BC_S3_ALLFACES_EXACT)
    assert torch.equal(vals, v2)
    plan.close()


def test_s3_hexahedron_kernel_falls_back_on_per_gauss_point_material(env):
    """Its premise (one (E, nu) per element) is tested bitwise on the records: a material that varies inside an element
    drops the plan to the general kernel, whose K is then the result."""
    O = env["O"]
    dims = (9, 8, 7)
    coords = O.perturb_coords(O.uniform_coords(3, *dims), 3, *dims)
    case = _s3_case_from_coords(env, dims, coords, vary_material=True)
    plan, vals, b = gpu_assemble(env, case)
    assert plan.fast_path() == 0
    assert rel(vals, case["K"]) <= TOL and rel(b, case["b"]) <= TOL
    plan.close()


@pytest.mark.parametrize("kind", ["p3", "s3", "e2"])
def test_general_geometry_drops_to_the_general_kernel(env, kind):
    case = make_case(env, kind, (70, 40) if kind == "e2" else (8, 7, 9), perturb=True)
    plan, vals, b = gpu_assemble(env, case)
    assert plan.fast_path() == (1 if kind == "s3" else 0)  # S3: hexahedra with one material each stay at level 1
    assert rel(vals, case["K"]) <= TOL
    plan.close()


def test_one_moved_interior_node_is_enough_to_fall_back(env):
    O = env["O"]
    dims = (9, 9, 9)
    c = O.uniform_coords(3, *dims).reshape(-1, 3)
    c[4 + 9 * (4 + 9 * 4)] += [0.01, -0.02, 0.015]
    case = _s3_case_from_coords(env, dims, c.reshape(-1).copy())
    plan, vals, b = gpu_assemble(env, case)
    assert plan.fast_path() == 1
    assert rel(vals, case["K"]) <= TOL
    plan.close()


def test_material_varying_inside_an_element_falls_back(env):
    O = env["O"]
    dims = (7, 6, 8)
    case = _s3_case_from_coords(env, dims, O.uniform_coords(3, *dims), vary_material=True)
    plan, vals, b = gpu_assemble(env, case)
    assert plan.fast_path() == 0
    assert rel(vals, case["K"]) <= TOL
    plan.close()


@pytest.mark.parametrize("kind", ["p3", "s3", "e2"])
@pytest.mark.parametrize("level", [0, 1, 2])
def test_all_kernel_levels_agree_with_the_oracle_on_a_uniform_mesh(env, kind, level):
    torch, F, dev = env["torch"], env["F"], env["dev"]
    case = make_case(env, kind, (150, 71) if kind == "e2" else (20, 9, 19), perturb=False)
    M, N, P = case["dims"]
    plan = F.Plan(case["problem"], M, N, P, coords=torch.from_numpy(case["coords"]).to(dev))
    plan.set_fast_path(level)
    vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
    mat = None if case["props"] is None else torch.from_numpy(case["props"]).to(dev)
    plan.assemble_system(vals, None, material=mat, stride=case["stride"], bc=case["bc"])
    assert plan.fast_path() == level
    assert rel(vals.cpu().numpy(), case["K"]) <= TOL
    plan.close()


def test_e2_material_varying_inside_an_element_falls_back(env):
    case = make_case(env, "e2", (90, 37), perturb=False, vary_material=True)
    plan, vals, b = gpu_assemble(env, case)
    assert plan.fast_path() == 0
    assert rel(vals, case["K"]) <= TOL and rel(b, case["b"]) <= TOL
    plan.close()


def test_e2_rectangles_keep_the_row_sweep(env):
    case = make_case(env, "e2", (131, 67), perturb=False)  # two x tiles, ragged; three y chunks
    plan, vals, b = gpu_assemble(env, case)
    assert plan.fast_path() == 2
    assert rel(vals, case["K"]) <= TOL and rel(b, case["b"]) <= TOL
    plan.close()


@pytest.mark.parametrize("kind,dims", [("p3", (9, 8, 10)), ("e2", (70, 33)), ("s3", (8, 7, 9))])
@pytest.mark.parametrize("perturb", [False, True])
def test_solution_of_the_assembled_system_matches_the_oracle_system(env, kind, dims, perturb):
    """north_star: the KSP solution matches within 1e-12 relative.  KSP itself is out of scope; here both systems
    (GPU-assembled and oracle-assembled) are solved by the same sparse direct solver."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    case = make_case(env, kind, dims, perturb=perturb)
    plan, vals, b = gpu_assemble(env, case)
    n = len(case["rp"]) - 1
    Kg = sp.csr_matrix((vals, case["ci"], case["rp"]), shape=(n, n)).tocsc()
    Ko = sp.csr_matrix((case["K"], case["ci"], case["rp"]), shape=(n, n)).tocsc()
    ug, uo = spla.spsolve(Kg, b), spla.spsolve(Ko, case["b"])
    assert np.isfinite(uo).all() and np.linalg.norm(uo) > 0
    assert rel(ug, uo) <= 1e-12  # the north_star's tolerance
    plan.close()


# ---- N3: error integrals (DMDAIntegrateErrors3D) ----
@pytest.mark.parametrize("kind,dims", [("p3", (9, 7, 8)), ("s3", (6, 8, 7))])
@pytest.mark.parametrize("host", [False, True])
def test_error_integrals_match_the_oracle(env, kind, dims, host):
    torch, F, dev, O = env["torch"], env["F"], env["dev"], env["O"]
    M, N, P = dims
    dof = 1 if kind == "p3" else 3
    coords = O.perturb_coords(O.uniform_coords(3, M, N, P), 3, M, N, P)
    rng = np.random.default_rng(11)
    u = rng.uniform(-1, 1, size=M * N * P * dof)
    h, l2, h1 = O.errors(kind, M, N, P, coords, u)
    plan = F.Plan(F.POISSON3D_Q1 if kind == "p3" else F.ELASTIC3D_Q1, M, N, P, coords=torch.from_numpy(coords).to(dev))
    s0, s1 = plan.error_integrals(u if host else torch.from_numpy(u).to(dev))
    assert abs(np.sqrt(s0) - l2) <= 1e-12 * l2 and abs(np.sqrt(s1) - h1) <= 1e-12 * h1
    assert h == 1.0 / (M - 1)
    plan.close()


def test_error_integrals_add_up_over_slabs(env):
    torch, F, dev, O, slab = env["torch"], env["F"], env["dev"], env["O"], env["slab"]
    M, N, P = 6, 5, 11
    coords = O.perturb_coords(O.uniform_coords(3, M, N, P), 3, M, N, P)
    u = np.random.default_rng(5).uniform(-1, 1, size=M * N * P * 3)
    _, l2, h1 = O.errors("s3", M, N, P, coords, u)
    tot = np.zeros(2)
    for (ks, ke) in slab.slab_ranges(P, 3):
        cks, cke = max(ks - 1, 0), min(ke + 1, P)
        co = torch.from_numpy(coords.reshape(P, -1)[cks:cke].reshape(-1).copy()).to(dev)
        ul = torch.from_numpy(u.reshape(P, -1)[cks:cke].reshape(-1).copy()).to(dev)
        plan = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=co, ks=ks, ke=ke, cks=cks, cke=cke)
        tot += plan.error_integrals(ul)  # the MPI_Allreduce of S3:1137-1138
        plan.close()
    assert abs(np.sqrt(tot[0]) - l2) <= 1e-12 * l2 and abs(np.sqrt(tot[1]) - h1) <= 1e-12 * h1


# ---- N1: GaussPointCoefficients fill on device ----
@pytest.mark.parametrize("stride", [400, 416])
@pytest.mark.parametrize("host", [False, True])
def test_fill_properties_s3_matches_the_driver_fill(env, stride, host):
    torch, F, dev, O = env["torch"], env["F"], env["dev"], env["O"]
    M, N, P = 6, 5, 7
    coords = O.perturb_coords(O.uniform_coords(3, M, N, P), 3, M, N, P)
    ref = O.props_s3(M - 1, N - 1, P - 1, coords, E=90.0, nu=0.28, stride=stride).reshape(-1, stride)
    plan = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=torch.from_numpy(coords).to(dev))
    if host:
        out = np.full(ref.size, np.nan)
        plan.fill_properties(out, 90.0, 0.28, stride)
    else:
        t = torch.full((ref.size,), float("nan"), dtype=torch.float64, device=dev)
        plan.fill_properties(t, 90.0, 0.28, stride)
        out = t.cpu().numpy()
    out = out.reshape(-1, stride)
    assert not np.isnan(out).any()
    assert np.array_equal(out[:, 96:160], ref[:, 96:160])    # E, nu, E_B, nu_B: copies
    assert np.array_equal(out[:, 400:], ref[:, 400:])        # R3 tail: zeros
    for lo, hi in ((0, 96), (160, 184), (184, 400)):         # positions; body force; displacement gradient
        assert rel(out[:, lo:hi], ref[:, lo:hi]) <= TOL
    # and the system assembled from the device-filled array is the oracle's system
    rp, ci = O.pattern(3, 3, M, N, P)
    K, b = O.assemble_elastic3d(M, N, P, coords, ref.reshape(-1).copy() if stride == 400 else
                                O.props_s3(M - 1, N - 1, P - 1, coords), rp, ci)
    O.bc_s3(M, N, P, coords, rp, ci, K, b)
    vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
    bb = torch.empty(plan.nrows, dtype=torch.float64, device=dev)
    plan.assemble_system(vals, bb, material=torch.from_numpy(out.reshape(-1).copy()).to(dev), stride=stride,
                         bc=F.BC_S3_ALLFACES_EXACT)
    assert rel(vals.cpu().numpy(), K) <= TOL and rel(bb.cpu().numpy(), b) <= TOL
    plan.close()


def test_fill_properties_e2_matches_the_driver_fill(env):
    torch, F, dev, O = env["torch"], env["F"], env["dev"], env["O"]
    M, N = 23, 11
    coords = O.perturb_coords(O.uniform_coords(2, M, N), 2, M, N)
    ref = O.props_e2(M - 1, N - 1, coords, E=1.0, nu=0.33).reshape(-1, 24)
    plan = F.Plan(F.PLANE_STRAIN2D_Q1, M, N, 1, coords=torch.from_numpy(coords).to(dev))
    t = torch.full((ref.size,), float("nan"), dtype=torch.float64, device=dev)
    plan.fill_properties(t, 1.0, 0.33, 24)
    out = t.cpu().numpy().reshape(-1, 24)
    assert np.array_equal(out[:, 8:], ref[:, 8:]) and rel(out[:, :8], ref[:, :8]) <= TOL
    plan.close()


def test_fill_properties_of_a_slab_uses_its_element_layers(env):
    torch, F, dev, O, slab = env["torch"], env["F"], env["dev"], env["O"], env["slab"]
    M, N, P = 5, 4, 9
    coords = O.perturb_coords(O.uniform_coords(3, M, N, P), 3, M, N, P)
    ref = O.props_s3(M - 1, N - 1, P - 1, coords).reshape(P - 1, -1)
    for (ks, ke) in slab.slab_ranges(P, 3):
        cks, cke = max(ks - 1, 0), min(ke + 1, P)
        els, ele = slab.element_range(ks, ke)
        co = torch.from_numpy(coords.reshape(P, -1)[cks:cke].reshape(-1).copy()).to(dev)
        plan = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=co, ks=ks, ke=ke, cks=cks, cke=cke)
        t = torch.empty((ele - els) * ref.shape[1], dtype=torch.float64, device=dev)
        plan.fill_properties(t, 90.0, 0.28, 400)
        assert rel(t.cpu().numpy(), ref[els:ele].reshape(-1)) <= TOL
        plan.close()


# ---- N4: matrix-free operator apply ----
@pytest.mark.parametrize("kind,dims", [("p3", (7, 6, 8)), ("e2", (31, 17)), ("s3", (6, 7, 5))])
def test_operator_apply_equals_the_assembled_matrix_times_x(env, kind, dims):
    import scipy.sparse as sp
    torch, F, dev = env["torch"], env["F"], env["dev"]
    case = make_case(env, kind, dims, perturb=True, vary_material=(kind != "p3"))
    M, N, P = case["dims"]
    n = len(case["rp"]) - 1
    K = sp.csr_matrix((case["K"], case["ci"], case["rp"]), shape=(n, n))
    x = np.random.default_rng(2).uniform(-1, 1, size=n)
    plan = F.Plan(case["problem"], M, N, P, coords=torch.from_numpy(case["coords"]).to(dev))
    y = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
    mat = None if case["props"] is None else torch.from_numpy(case["props"]).to(dev)
    plan.apply_operator(torch.from_numpy(x).to(dev), y, material=mat, stride=case["stride"], bc=case["bc"])
    assert rel(y.cpu().numpy(), K @ x) <= 1e-12
    plan.close()


def test_operator_apply_over_slabs_with_the_interface_exchange(env):
    import scipy.sparse as sp
    torch, F, dev, slab = env["torch"], env["F"], env["dev"], env["slab"]
    case = make_case(env, "s3", (5, 4, 10), perturb=True, vary_material=True)
    M, N, P = case["dims"]
    n = len(case["rp"]) - 1
    K = sp.csr_matrix((case["K"], case["ci"], case["rp"]), shape=(n, n))
    x = np.random.default_rng(4).uniform(-1, 1, size=n)
    plane_e, plane_n = (M - 1) * (N - 1), M * N * 3
    ranges = slab.slab_ranges(P, 3)
    plans, ys, bufs = [], [], []
    for (ks, ke) in ranges:
        cks, cke = max(ks - 1, 0), min(ke + 1, P)
        els, ele = slab.element_range(ks, ke)
        co = torch.from_numpy(case["coords"].reshape(P, -1)[cks:cke].reshape(-1).copy()).to(dev)
        xl = torch.from_numpy(x.reshape(P, -1)[cks:cke].reshape(-1).copy()).to(dev)
        mat = torch.from_numpy(case["props"].reshape(-1, plane_e * 400)[els:ele].reshape(-1).copy()).to(dev)
        plan = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=co, ks=ks, ke=ke, cks=cks, cke=cke)
        (_, sc), (_, rc) = plan.interface_buffers()
        send = torch.zeros(max(sc, 1), dtype=torch.float64, device=dev)
        recv = torch.zeros(max(rc, 1), dtype=torch.float64, device=dev)
        plan.interface_set_buffers(send if sc else None, recv if rc else None)
        y = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
        plan.apply_operator(xl, y, material=mat, stride=400, bc=case["bc"], e_ks=els)
        plans.append(plan); ys.append(y); bufs.append((send, recv, sc, rc))
    for r in range(len(ranges) - 1):
        bufs[r][1][:bufs[r][3]].copy_(bufs[r + 1][0][:bufs[r + 1][2]])
    for r in range(len(ranges)):
        plans[r].interface_apply(None, ys[r], bc=case["bc"])
    y = np.concatenate([t.cpu().numpy() for t in ys])
    assert rel(y, K @ x) <= 1e-12
    for p in plans:
        p.close()


def _fixture_names():
    import os
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "systems.npz"))
    return sorted({k.split("/")[0] for k in f.files})


@pytest.mark.parametrize("name", _fixture_names())
def test_committed_system_fixtures(env, name):
    """The HIP path against tests/golden/systems.npz (no oracle call: the expected K and b are data)."""
    import os
    torch, F, dev = env["torch"], env["F"], env["dev"]
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "systems.npz"))
    kind, M, N, P, stride = (int(v) for v in f[name + "/meta"])
    problem = [F.POISSON3D_Q1, F.PLANE_STRAIN2D_Q1, F.ELASTIC3D_Q1][kind]
    bc = [F.BC_POISSON_INLOOP, F.BC_E2_COMPRESSION, F.BC_S3_ALLFACES_EXACT][kind]
    case = dict(problem=problem, dims=(M, N, P), coords=f[name + "/coords"],
                props=f[name + "/props"] if stride else None, stride=stride, bc=bc)
    plan, vals, b = gpu_assemble(env, case)
    assert rel(vals, f[name + "/K"]) <= TOL and rel(b, f[name + "/b"]) <= TOL
    plan.close()


# --------------------------------------------------------------------------
# N2: COO landing; N4 second half: rigid-body near-nullspace
# --------------------------------------------------------------------------
@pytest.mark.parametrize("kind,dims,dtype", [("p3", (5, 4, 6), np.int32), ("s3", (4, 5, 7), np.int64), ("e2", (9, 6), np.int32)])
def test_coo_indices_address_the_csr_value_slots(env, kind, dims, dtype):
    """MatSetValuesCOO(A, vals) with these indices must build the same matrix as the CSR arrays: bit-exact
    integer work, single GPU and per slab (global row ids)."""
    torch, F, dev, slab = env["torch"], env["F"], env["dev"], env["slab"]
    case = make_case(env, kind, dims)
    M, N, P = case["dims"]
    L = P if kind != "e2" else N
    dof = {"p3": 1, "e2": 2, "s3": 3}[kind]
    plane = M * N if kind != "e2" else M
    expect_i = np.repeat(np.arange(len(case["rp"]) - 1), np.diff(case["rp"]))
    for ks, ke in [(0, L)] + list(slab.slab_ranges(L, 2)):
        cks, cke = max(ks - 1, 0), min(ke + 1, L)
        cl = case["coords"].reshape(L, -1)[cks:cke].reshape(-1)
        plan = F.Plan(case["problem"], M, N, P, coords=torch.from_numpy(cl).to(dev), ks=ks, ke=ke, cks=cks, cke=cke)
        ci_ = torch.empty(plan.nnz, dtype=torch.int32 if dtype == np.int32 else torch.int64, device=dev)
        cj_ = torch.empty_like(ci_)
        plan.coo_indices(ci_, cj_)
        r0, r1 = plane * ks * dof, plane * ke * dof
        a, b = case["rp"][r0], case["rp"][r1]
        assert np.array_equal(ci_.cpu().numpy(), expect_i[a:b]) and np.array_equal(cj_.cpu().numpy(), case["ci"][a:b])
        hi, hj = np.empty(plan.nnz, dtype=dtype), np.empty(plan.nnz, dtype=dtype)  # host outputs
        plan.coo_indices(hi, hj)
        assert np.array_equal(hi, expect_i[a:b]) and np.array_equal(hj, case["ci"][a:b])
        plan.close()


def _petsc_rigid_body(coords, nsd):
    """MatNullSpaceCreateRigidBody as PETSc documents it (src/mat/interface/matnull.c; PETSc is not part of the
    reference tree -- parity unpinned): raw vectors, then classical Gram-Schmidt + normalise of the rotations."""
    x = coords.reshape(-1, nsd)
    n = len(x)
    sN = 1.0 / np.sqrt(n)
    nm = 6 if nsd == 3 else 3
    v = np.zeros((nm, n, nsd))
    for k in range(nsd):
        v[k, :, k] = sN
    if nsd == 3:
        v[3, :, 0], v[3, :, 1] = x[:, 1], -x[:, 0]
        v[4, :, 1], v[4, :, 2] = -x[:, 2], x[:, 1]
        v[5, :, 0], v[5, :, 2] = x[:, 2], -x[:, 0]
    else:
        v[2, :, 0], v[2, :, 1] = -x[:, 1], x[:, 0]
    v = v.reshape(nm, -1)
    for i in range(nsd, nm):
        dots = v[:i] @ v[i]
        v[i] = v[i] - dots @ v[:i]
        v[i] /= np.linalg.norm(v[i])
    return v


@pytest.mark.parametrize("kind,dims", [("s3", (5, 4, 6)), ("e2", (7, 9))])
def test_rigid_body_modes(env, kind, dims):
    """Orthonormal, equal to PETSc's construction, in the kernel of the unconstrained K (any mesh), and the same
    whether built on one GPU or slab by slab from the summed moments."""
    torch, F, dev, slab, O = env["torch"], env["F"], env["dev"], env["slab"], env["O"]
    case = make_case(env, kind, dims, perturb=True)
    M, N, P = case["dims"]
    nsd = 3 if kind == "s3" else 2
    L = P if nsd == 3 else N
    nm = 6 if nsd == 3 else 3
    ref = _petsc_rigid_body(case["coords"], nsd)
    plan = F.Plan(case["problem"], M, N, P, coords=torch.from_numpy(case["coords"]).to(dev))
    mom = plan.coord_moments()
    x = case["coords"].reshape(-1, nsd)
    assert mom[0] == len(x) and abs(mom[1] - x[:, 0].sum()) < 1e-12 * len(x)
    modes = torch.empty(nm * plan.nrows, dtype=torch.float64, device=dev)
    plan.rigid_body_modes(mom, modes)
    Q = modes.cpu().numpy().reshape(nm, -1)
    assert np.abs(Q @ Q.T - np.eye(nm)).max() < 1e-12
    assert np.abs(Q - ref).max() < 1e-12
    # K (no Dirichlet rows) annihilates them
    vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
    mat = torch.from_numpy(case["props"]).to(dev)
    plan.assemble_system(vals, None, material=mat, stride=case["stride"], bc=F.BC_NONE)
    import scipy.sparse as sp
    K = sp.csr_matrix((vals.cpu().numpy(), case["ci"].astype(np.int64), case["rp"]), shape=(plan.nrows, plan.nrows))
    assert np.abs(K @ Q.T).max() < 1e-10 * np.abs(vals.cpu().numpy()).max()
    plan.close()
    # slab by slab, host output
    parts, moms, plans = [], [], []
    for ks, ke in slab.slab_ranges(L, 3):
        cks, cke = max(ks - 1, 0), min(ke + 1, L)
        cl = case["coords"].reshape(L, -1)[cks:cke].reshape(-1)
        pl = F.Plan(case["problem"], M, N, P, coords=torch.from_numpy(cl).to(dev), ks=ks, ke=ke, cks=cks, cke=cke)
        plans.append(pl)
        moms.append(pl.coord_moments())
    tot = np.sum(moms, axis=0)
    for pl in plans:
        out = np.empty(nm * pl.nrows)
        pl.rigid_body_modes(tot, out)
        parts.append(out.reshape(nm, -1))
        pl.close()
    assert np.abs(np.concatenate(parts, axis=1) - ref).max() < 1e-12


@pytest.mark.parametrize("stride,shift", [(401, 0), (400, 1), (402, 0)])
def test_elastic_rhs_odd_stride_or_base_takes_the_per_element_kernel(env, stride, shift):
    """The streaming RHS kernel reads 16-byte runs; an odd stride or an 8-byte-aligned base must still give the
    same b (host routes them to the thread-per-element kernel)."""
    torch, F, dev, O = env["torch"], env["F"], env["dev"], env["O"]
    case = make_case(env, "s3", (6, 5, 4), vary_material=True)
    M, N, P = case["dims"]
    nel = (M - 1) * (N - 1) * (P - 1)
    wide = np.zeros((nel, stride))
    wide[:, :400] = case["props"].reshape(nel, 400)
    buf = torch.zeros(nel * stride + shift, dtype=torch.float64, device=dev)
    buf[shift:] = torch.from_numpy(wide.reshape(-1)).to(dev)
    plan = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=torch.from_numpy(case["coords"]).to(dev))
    vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
    b = torch.empty(plan.nrows, dtype=torch.float64, device=dev)
    plan.assemble_system(vals, b, material=buf[shift:], stride=stride, bc=F.BC_S3_ALLFACES_EXACT)
    assert rel(vals.cpu().numpy(), case["K"]) <= TOL and rel(b.cpu().numpy(), case["b"]) <= TOL
    plan.close()


def test_elastic_uniform_material_rhs_is_the_dirichlet_data(env):
    """FEMASM_MAT_UNIFORM carries no loads: b holds only the Dirichlet values (zero body force, no traction)."""
    torch, F, dev, O = env["torch"], env["F"], env["dev"], env["O"]
    M, N, P = 5, 6, 4
    coords = O.uniform_coords(3, M, N, P)
    rp, ci = O.pattern(3, 3, M, N, P)
    props = O.props_s3(M - 1, N - 1, P - 1, coords).reshape(-1, 400)
    props[:, 160:400] = 0.0  # fx,fy,fz,gx,gy,gz
    K, bref = O.assemble_elastic3d(M, N, P, coords, props.reshape(-1), rp, ci)
    O.bc_s3(M, N, P, coords, rp, ci, K, bref)
    plan = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=torch.from_numpy(coords).to(dev))
    b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
    plan.assemble_system(None, b, material=None, E=90.0, nu=0.28, bc=F.BC_S3_ALLFACES_EXACT)
    assert np.abs(b.cpu().numpy() - bref).max() <= 1e-15
    plan.close()


@pytest.mark.parametrize("host_list", [False, True])
def test_zero_rows_list_reproduces_the_builtin_dirichlet_kinds(env, host_list):
    """femasm_zero_rows on a BC_NONE assembly with the list of Dirichlet rows and their values = the built-in
    S3 all-faces kind (MatZeroRows + VecSetValues semantics), on one GPU and per slab (foreign rows skipped)."""
    torch, F, dev, slab, O = env["torch"], env["F"], env["dev"], env["slab"], env["O"]
    case = make_case(env, "s3", (6, 5, 7), vary_material=True)
    M, N, P = case["dims"]
    flags = O.bc_flags("s3", M, N, P).astype(bool)
    rows = np.nonzero(flags)[0].astype(np.int64)
    rows = np.concatenate([rows, rows[:7]])  # duplicates, as the drivers' face loops produce on edges
    vals = case["b"][rows].copy()            # after the oracle's BC step b holds the Dirichlet values there
    for ks, ke in [(0, P)] + list(slab.slab_ranges(P, 2)):
        cks, cke, els = max(ks - 1, 0), min(ke + 1, P), max(ks - 1, 0)
        cl = case["coords"].reshape(P, -1)[cks:cke].reshape(-1)
        ml = case["props"].reshape(P - 1, -1)[els:ke - 1].reshape(-1)
        outs = []
        for kind in ("list", "builtin"):
            plan = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=torch.from_numpy(cl).to(dev), ks=ks, ke=ke, cks=cks, cke=cke)
            v = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
            b = torch.empty(plan.nrows, dtype=torch.float64, device=dev)
            mat = torch.from_numpy(ml).to(dev)
            if kind == "builtin":
                plan.assemble_system(v, b, material=mat, stride=400, bc=F.BC_S3_ALLFACES_EXACT, e_ks=els)
            else:
                plan.assemble_system(v, b, material=mat, stride=400, bc=F.BC_NONE, e_ks=els)
                r_, v_ = (rows, vals) if host_list else (torch.from_numpy(rows).to(dev), torch.from_numpy(vals).to(dev))
                plan.zero_rows(r_, v_, v, b)
            outs.append((v.cpu().numpy(), b.cpu().numpy()))
            plan.close()
        # the top layer of a lower slab is partial in both (same partial sums); Dirichlet rows identical
        assert rel(outs[0][0], outs[1][0]) <= TOL and rel(outs[0][1], outs[1][1]) <= TOL
        lo, hi = 3 * M * N * ks, 3 * M * N * ke
        own = flags[lo:hi]
        assert np.array_equal(outs[0][1][own], case["b"][lo:hi][own])
