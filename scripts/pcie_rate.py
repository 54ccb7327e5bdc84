"""PCIe-inclusive rate of the boundary: the driver hands in HOST pointers (MatSeqAIJGetArray / VecGetArray)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from petsc_amd import femasm as F, synth
dev = torch.device("cuda:0")
nel = 128; M = nel + 1
coords = synth.uniform_coords(3, M, M, M, 0, M, device=dev)
mat_d = synth.props_s3(M, M, M, coords, 0, nel, ck0=0, E=90.0, nu=0.28)
coords_h, mat_h = coords.cpu().numpy(), mat_d.cpu().numpy()
plan = F.Plan(F.ELASTIC3D_Q1, M, M, M, coords=coords_h)
vals, b = np.empty(plan.nnz), np.empty(plan.nrows)
for it in range(3):
    t0 = time.perf_counter()
    st = plan.assemble_system(vals, b, material=mat_h, stride=400, bc=F.BC_S3_ALLFACES_EXACT, want_stats=True)
    dt = time.perf_counter() - t0
    print(f"host pointers: {1e3*dt:.1f} ms per call ({nel**3/dt:.3g} elements/s), staging {st.ms_staging:.1f} ms, "
          f"K kernels {st.ms_kernel_K:.2f} ms, b kernels {st.ms_kernel_rhs:.2f} ms; "
          f"bytes over PCIe: in {mat_h.nbytes/1e9:.2f} GB, out {(vals.nbytes+b.nbytes)/1e9:.2f} GB")
plan.close()
