"""Cost of the multi-GPU extras on one slab (rank 1 of 2, 128x128x256 elastic): femasm_assemble_interface and
femasm_interface_apply, timed with events on the plan's stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from petsc_amd import femasm as F, slab, synth

dev = torch.device("cuda:0")
nel, world, rank = 128, 2, int(sys.argv[1]) if len(sys.argv) > 1 else 1
M = N = nel + 1
P = nel * world + 1
ks, ke = slab.slab_ranges(P, world)[rank]
cks, cke = max(ks - 1, 0), min(ke + 1, P)
els, ele = max(ks - 1, 0), ke - 1
coords = synth.uniform_coords(3, M, N, P, cks, cke, device=dev)
mat = synth.props_s3(M, N, P, coords, els, ele, ck0=cks, E=90.0, nu=0.28)
plan = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=coords, ks=ks, ke=ke, cks=cks, cke=cke)
plan.set_stream(torch.cuda.current_stream().cuda_stream)
vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
b = torch.empty(plan.nrows, dtype=torch.float64, device=dev)
(_, sc), (_, rc) = plan.interface_buffers()
send = torch.zeros(max(sc, 1), dtype=torch.float64, device=dev)
recv = torch.zeros(max(rc, 1), dtype=torch.float64, device=dev)
plan.interface_set_buffers(send if sc else None, recv if rc else None)
bc = F.BC_S3_ALLFACES_EXACT
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
t_if = timed(lambda: plan.assemble_interface(material=mat, stride=400, bc=bc, e_ks=els))
t_sys = timed(lambda: plan.assemble_system(vals, b, material=mat, stride=400, bc=bc, e_ks=els))
def both():
    plan.assemble_interface(material=mat, stride=400, bc=bc, e_ks=els)
    plan.assemble_system(vals, b, material=mat, stride=400, bc=bc, e_ks=els)
t_both = timed(both)
t_ap = timed(lambda: plan.interface_apply(vals, b, bc=bc))
print(f"rank {rank}/{world}: send {sc*8/1e6:.1f} MB recv {rc*8/1e6:.1f} MB | assemble_interface {t_if:.3f} ms | "
      f"assemble_system (incl. interface pass) {t_sys:.3f} ms | split pair {t_both:.3f} ms | interface_apply {t_ap:.3f} ms")
