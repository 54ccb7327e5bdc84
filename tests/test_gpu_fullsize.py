"""GPU tests at BASELINE.json's full sizes (-m gpu).  The oracle cannot assemble these in seconds, so the
checks are (a) a window of interior rows against an oracle assembly of the same sub-box, and (b)
size-independent properties: rigid translations in the kernel of K on non-Dirichlet rows, unit Dirichlet
rows, no unwritten slot, bitwise-equal repeat."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(O):
    import torch
    from petsc_amd import femasm as F
    from petsc_amd import synth
    assert torch.cuda.is_available()
    return dict(torch=torch, F=F, O=O, synth=synth, dev=torch.device("cuda:0"))


def _window_rows(rp, dof, M, N, i0, j0, k0, n):
    """rows of the strictly interior nodes of the n^3-node sub-box at (i0,j0,k0)"""
    rows = []
    for k in range(k0 + 1, k0 + n - 1):
        for j in range(j0 + 1, j0 + n - 1):
            for i in range(i0 + 1, i0 + n - 1):
                rows += [dof * (i + M * (j + N * k)) + c for c in range(dof)]
    return np.array(rows)


def test_elastic3d_128_window_and_properties(env):
    torch, F, O, synth, dev = env["torch"], env["F"], env["O"], env["synth"], env["dev"]
    nel = 128
    M = nel + 1
    coords = synth.perturb_coords(synth.uniform_coords(3, M, M, M, device=dev), 3, M, M, M)
    props = synth.props_s3(M, M, M, coords)
    plan = F.Plan(F.ELASTIC3D_Q1, M, M, M, coords=coords)
    vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
    b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
    plan.assemble_system(vals, b, material=props, stride=400, bc=F.BC_S3_ALLFACES_EXACT)
    assert not torch.isnan(vals).any() and not torch.isnan(b).any()
    assert plan.nnz == 513599625 and plan.nrows == 6440067

    # (a) window: 7^3-node sub-box at (40,50,60) vs the oracle on the same coordinates/material
    n, (i0, j0, k0) = 7, (40, 50, 60)
    c4 = coords.reshape(M, M, M, 3)[k0:k0 + n, j0:j0 + n, i0:i0 + n].contiguous().cpu().numpy().reshape(-1)
    p4 = props.reshape(nel, nel, nel, 400)[k0:k0 + n - 1, j0:j0 + n - 1, i0:i0 + n - 1].contiguous().cpu().numpy().reshape(-1)
    rps, cis = O.pattern(3, 3, n, n, n)
    Ks, bs = O.assemble_elastic3d(n, n, n, c4, p4, rps, cis)
    rows_g = _window_rows(None, 3, M, M, i0, j0, k0, n)
    rows_s = _window_rows(None, 3, n, n, 0, 0, 0, n)
    crow = torch.empty(plan.nrows + 1, dtype=torch.int64)
    rp, _ = plan.pattern()
    num = den = 0.0
    vh = None
    for rg, rs in zip(rows_g, rows_s):
        g = vals[rp[rg]:rp[rg + 1]].cpu().numpy()
        s = Ks[rps[rs]:rps[rs + 1]]
        assert g.shape == s.shape == (81,)
        num += np.sum((g - s) ** 2); den += np.sum(s ** 2)
    assert np.sqrt(num / den) <= 1e-13
    bg = b.cpu().numpy()[rows_g]
    assert np.linalg.norm(bg - bs[rows_s]) <= 1e-13 * np.linalg.norm(bs[rows_s])

    # (b) properties on the whole 4.1 GB matrix: K * (rigid translation) = 0 on non-Dirichlet rows
    crow = torch.from_numpy(rp).to(dev)
    _, ci = plan.pattern()
    col = torch.from_numpy(ci.astype(np.int64)).to(dev)
    A = torch.sparse_csr_tensor(crow, col, vals, size=(plan.nrows, plan.nrows))
    idx = torch.arange(M ** 3, device=dev)
    i, j, k = idx % M, (idx // M) % M, idx // (M * M)
    bnd = ((i == 0) | (i == M - 1) | (j == 0) | (j == M - 1) | (k == 0) | (k == M - 1)).repeat_interleave(3)
    scale = vals.abs().max()
    for c in range(3):
        t = torch.zeros(plan.nrows, dtype=torch.float64, device=dev)
        t[c::3] = 1.0
        y = A @ t
        assert y[~bnd].abs().max() <= 1e-11 * scale
        assert torch.equal(y[bnd], t[bnd])  # unit Dirichlet rows
    del A
    vals2 = torch.empty_like(vals)
    plan.assemble_system(vals2, None, material=props, stride=400, bc=F.BC_S3_ALLFACES_EXACT)
    assert torch.equal(vals, vals2)
    plan.close()


def test_poisson3d_256_and_elastic2d_2048_properties(env):
    torch, F, O, synth, dev = env["torch"], env["F"], env["O"], env["synth"], env["dev"]
    # P3 256^3: boundary rows = valence * identity, interior row sums = 0, b = 0 on the boundary
    nel = 256
    M = nel + 1
    coords = synth.perturb_coords(synth.uniform_coords(3, M, M, M, device=dev), 3, M, M, M)
    plan = F.Plan(F.POISSON3D_Q1, M, M, M, coords=coords)
    vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
    b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
    plan.assemble_system(vals, b, bc=F.BC_POISSON_INLOOP)
    assert plan.nnz == 454756609 and not torch.isnan(vals).any() and not torch.isnan(b).any()
    rp, ci = plan.pattern()
    A = torch.sparse_csr_tensor(torch.from_numpy(rp).to(dev), torch.from_numpy(ci.astype(np.int64)).to(dev), vals,
                                size=(plan.nrows, plan.nrows))
    y = A @ torch.ones(plan.nrows, dtype=torch.float64, device=dev)
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
    # window check against the oracle
    n, (i0, j0, k0) = 6, (100, 7, 200)
    c4 = coords.reshape(M, M, M, 3)[k0:k0 + n, j0:j0 + n, i0:i0 + n].contiguous().cpu().numpy().reshape(-1)
    rps, cis = O.pattern(3, 1, n, n, n)
    Ks, bs = O.assemble_poisson3d(n, n, n, c4, rps, cis)
    for rg, rs in zip(_window_rows(None, 1, M, M, i0, j0, k0, n), _window_rows(None, 1, n, n, 0, 0, 0, n)):
        g = vals[rp[rg]:rp[rg + 1]].cpu().numpy()
        s = Ks[rps[rs]:rps[rs + 1]]
        assert np.linalg.norm(g - s) <= 1e-13 * np.linalg.norm(s)
        assert abs(b[rg].item() - bs[rs]) <= 1e-13 * abs(bs[rs])
    del A
    plan.close()

    # E2 2048^2 on the reference's uniform mesh: every interior element matrix is the golden one
    nel = 2048
    M = nel + 1
    coords = synth.uniform_coords(2, M, M, device=dev)
    props = synth.props_e2(M, M, coords)
    plan = F.Plan(F.PLANE_STRAIN2D_Q1, M, M, coords=coords)
    vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
    b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
    plan.assemble_system(vals, b, material=props, stride=24, bc=F.BC_E2_COMPRESSION)
    assert plan.nnz == 151044100 and not torch.isnan(vals).any()
    rp, ci = plan.pattern()
    # diagonal of an interior ux row = 4 * Ke[0] of the golden plane-strain element
    r = 2 * (1000 + M * 1000)
    row = vals[rp[r]:rp[r + 1]].cpu().numpy()
    diag = row[np.searchsorted(ci[rp[r]:rp[r + 1]], r)]
    assert abs(diag - 4 * 0.61919504643982903) <= 1e-12
    bh = b.cpu().numpy().reshape(M, M, 2)
    assert (bh[:, 0, 0] == 1.0).all() and (bh[:, -1, 0] == -1.0).all() and not bh[:, 1:-1].any()
    plan.close()


def _bench_ranks(extra, port, nproc=2, timeout=560):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(nproc), "--backend",
           "gloo", "--share-gpu", "--no-cpu-baseline", "--steps", "2", "--warmup", "1"] + extra
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


@pytest.mark.parametrize("workload,extra", [("elastic3d", ["--nel", "24"]), ("elastic3d", ["--nel", "32", "--strong"]),
                                            ("poisson3d", ["--nel", "40", "--strong"]),
                                            ("elastic3d", ["--nel", "20", "--perturb"])])
def test_bench_two_ranks_end_to_end_checks_values(workload, extra):
    """The N > 1 path of bench.py as the driver launches it (torch.distributed.run, one rank per process), rehearsed
    with two ranks sharing this one GPU over gloo: split interface call, non-blocking exchange, assembly, apply --
    and, with --check, every rank's rows against a one-plan assembly of the same global mesh."""
    d = _bench_ranks(["--workload", workload, "--check"] + extra, 29541)
    nel = int(extra[1])
    strong = "--strong" in extra
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["scaling"] == ("strong" if strong else "weak")
    assert d["config"]["global_elements"] == (1 if strong else 2) * nel ** 3
    assert d["roofline"]["kernel_level"] == ((1 if workload == "elastic3d" else 0) if "--perturb" in extra else 2)
    assert d["check"]["rel_K"] <= 1e-13 and d["check"]["rel_b"] <= 1e-13, d["check"]


def test_bench_two_ranks_strong_256_elastic_rehearsal():
    """north_star's scaling mesh (S3 256^3, strong) through the N-rank driver path, two ranks on this one GPU."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 150e9:
        pytest.skip("needs ~110 GB of HBM")
    d = _bench_ranks(["--workload", "elastic3d", "--nel", "256", "--strong"], 29543)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["global_elements"] == 256 ** 3
    assert d["roofline"]["kernel_level"] == 2 and d["value"] > 0


@pytest.mark.parametrize("kind", ["poisson3d", "elastic3d"])
def test_compiled_driver_on_the_c_abi_converges_at_second_order(kind):
    """examples/femasm_driver.cpp: the reference drivers' flow (mesh -> properties -> pattern -> assembly -> solve ->
    DMDAIntegrateErrors3D) in C++ on the C ABI alone, host arrays as in the unmodified drivers; exit code 0 means
    the L2 error of the manufactured solution fell 3-5x per halving of h (the reference's only known-answer check)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "femasm_driver")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe, kind, "4", "8", "16"], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("L2 error ratio") == 2


@pytest.mark.parametrize("kind,nel", [("poisson3d", 21), ("elastic3d", 14)])
def test_compiled_slab_driver_exchanges_over_rccl(kind, nel):
    """examples/femasm_slab_driver.cpp: two slabs on this GPU, femasm_assemble_interface -> femasm_interface_exchange
    (RCCL send / receive of the packed interface rows, a one-rank communicator) -> femasm_assemble_system ->
    femasm_interface_apply, from C++ on the C ABI alone; exit code 0 = the slabs equal the one-plan system to 1e-13."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "femasm_slab_driver")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe, kind, str(nel)], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "2 slabs over RCCL" in r.stdout


def test_build_then_smoke_in_one_process():
    """__graft_entry__.build() loads libfemasm.so, smoke() then uses torch: the library and torch must end up on
    one HIP runtime whatever the order (regression: the library loaded first saw no device)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("smoke: elastic3d") == 2
