"""RHS of the 128^3 S3 workload timed alone, after a K assembly, and after a large device memset (cache state test)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from petsc_amd import femasm as F, synth

dev = torch.device("cuda:0")
nel = 128
M = nel + 1
coords = synth.uniform_coords(3, M, M, M, 0, M, device=dev)
plan = F.Plan(F.ELASTIC3D_Q1, M, M, M, coords=coords)
mat = torch.empty(plan.nel * 400, dtype=torch.float64, device=dev)
plan.fill_properties(mat, 90.0, 0.28, 400)
vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
b = torch.empty(plan.nrows, dtype=torch.float64, device=dev)
bc = F.BC_S3_ALLFACES_EXACT
def rhs():
    return plan.assemble_system(None, b, material=mat, stride=400, bc=bc, want_stats=True)
def k():
    return plan.assemble_system(vals, None, material=mat, stride=400, bc=bc, want_stats=True)
for _ in range(3):
    rhs(); k()
t = [rhs().ms_kernel_rhs for _ in range(10)]
print("rhs after rhs      : %.3f ms" % (sum(t) / len(t)))
t = []
for _ in range(10):
    k(); t.append(rhs().ms_kernel_rhs)
print("rhs after K        : %.3f ms" % (sum(t) / len(t)))
big = torch.empty(1 << 29, dtype=torch.float64, device=dev)  # 4 GB
t = []
for _ in range(10):
    big.zero_(); torch.cuda.synchronize(); t.append(rhs().ms_kernel_rhs)
print("rhs after 4 GB fill: %.3f ms" % (sum(t) / len(t)))
t = []
for _ in range(10):
    rhs(); t.append(k().ms_kernel_K)
print("K after rhs        : %.3f ms" % (sum(t) / len(t)))
t = [k().ms_kernel_K for _ in range(10)]
print("K after K          : %.3f ms" % (sum(t) / len(t)))
