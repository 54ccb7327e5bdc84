"""Development check: all three problem kinds vs the oracle on small meshes (needs a GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from petsc_amd import femasm as F
from oracle import oracle as O

dev = torch.device("cuda:0")
def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)

def run(kind, dims, perturb):
    if kind == "p3":
        M, N, P = dims; nsd, dof = 3, 1
        coords = O.uniform_coords(3, M, N, P)
        if perturb: coords = O.perturb_coords(coords, 3, M, N, P)
        rp, ci = O.pattern(3, 1, M, N, P)
        Kref, bref = O.assemble_poisson3d(M, N, P, coords, rp, ci)
        plan = F.Plan(F.POISSON3D_Q1, M, N, P, coords=torch.from_numpy(coords).to(dev))
        mat, stride, bc = None, 0, F.BC_POISSON_INLOOP
    elif kind == "e2":
        M, N = dims; P = 1
        coords = O.uniform_coords(2, M, N)
        if perturb: coords = O.perturb_coords(coords, 2, M, N)
        rp, ci = O.pattern(2, 2, M, N)
        props = O.props_e2(M - 1, N - 1, coords)
        props[16:24] = 0.3  # non-zero body force in one element to exercise the RHS
        Kref, bref = O.assemble_elastic2d(M, N, coords, props, rp, ci)
        O.bc_e2(M, N, rp, ci, Kref, bref)
        plan = F.Plan(F.PLANE_STRAIN2D_Q1, M, N, coords=torch.from_numpy(coords).to(dev))
        mat, stride, bc = torch.from_numpy(props).to(dev), 24, F.BC_E2_COMPRESSION
    else:
        M, N, P = dims
        coords = O.uniform_coords(3, M, N, P)
        if perturb: coords = O.perturb_coords(coords, 3, M, N, P)
        rp, ci = O.pattern(3, 3, M, N, P)
        props = O.props_s3(M - 1, N - 1, P - 1, coords)
        Kref, bref = O.assemble_elastic3d(M, N, P, coords, props, rp, ci)
        O.bc_s3(M, N, P, coords, rp, ci, Kref, bref)
        plan = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=torch.from_numpy(coords).to(dev))
        mat, stride, bc = torch.from_numpy(props).to(dev), 400, F.BC_S3_ALLFACES_EXACT
    prp, pci = plan.pattern()
    okp = np.array_equal(prp, rp) and np.array_equal(pci, ci)
    vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
    b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
    st = plan.assemble_system(vals, b, material=mat, stride=stride, bc=bc, want_stats=True)
    v = vals.cpu().numpy(); bb = b.cpu().numpy()
    print(f"{kind} {dims} perturb={perturb}: pattern={okp} nan={np.isnan(v).sum()},{np.isnan(bb).sum()} "
          f"relK={rel(v, Kref):.3e} relb={rel(bb, bref):.3e} maxabsK={np.abs(v-Kref).max():.2e} "
          f"K_ms={st.ms_kernel_K:.3f} rhs_ms={st.ms_kernel_rhs:.3f}", flush=True)
    plan.close()

for perturb in (False, True):
    run("p3", (5, 5, 5), perturb)
    run("p3", (9, 6, 7), perturb)
    run("e2", (9, 7), perturb)
    run("e2", (40, 21), perturb)
    run("s3", (4, 4, 4), perturb)
    run("s3", (8, 6, 7), perturb)
run("p3", (33, 33, 33), True)
run("s3", (17, 17, 17), True)
print("done")
