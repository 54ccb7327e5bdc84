"""N4 timing: matrix-free y = K x against a CSR SpMV over the assembled values (128^3 elastic, one GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from petsc_amd import femasm as F, synth
dev = torch.device("cuda:0"); nel = 128; M = nel + 1
co = synth.uniform_coords(3, M, M, M, 0, M, device=dev)
co = synth.perturb_coords(co, 3, M, M, M, k0=0) if "--perturb" in sys.argv else co
plan = F.Plan(F.ELASTIC3D_Q1, M, M, M, coords=co)
plan.set_stream(torch.cuda.current_stream().cuda_stream)
mat = torch.empty(plan.nel * 400, dtype=torch.float64, device=dev); plan.fill_properties(mat, 90.0, 0.28, 400)
x = torch.rand(plan.nrows, dtype=torch.float64, device=dev); y = torch.empty_like(x)
bc = F.BC_S3_ALLFACES_EXACT
def timed(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
t_mf = timed(lambda: plan.apply_operator(x, y, material=mat, stride=400, bc=bc))
print(f"matrix-free apply {t_mf:.3f} ms", flush=True)
if "--no-spmv" in sys.argv:
    sys.exit(0)
vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev); b = torch.empty(plan.nrows, dtype=torch.float64, device=dev)
plan.assemble_system(vals, b, material=mat, stride=400, bc=bc)
rp_h, ci_h = plan.pattern()
rp = torch.from_numpy(rp_h).to(dev); ci = torch.from_numpy(ci_h.astype('int64')).to(dev)  # same index dtype for both
A = torch.sparse_csr_tensor(rp, ci, vals, size=(plan.nrows, plan.nrows))
y2 = A @ x
t_sp = timed(lambda: torch.mv(A, x) if False else A @ x)
err = float((y - y2).norm() / y2.norm())
print(f"matrix-free apply {t_mf:.3f} ms | CSR SpMV (rocSPARSE via torch, {vals.numel()*12/1e9:.1f} GB of values+indices) {t_sp:.3f} ms | rel. diff {err:.2e}")
