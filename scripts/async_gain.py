"""Step time of the headline workload: synchronous calls with and without stats against the asynchronous mode."""
import sys, time, torch
sys.path.insert(0, '.')
from petsc_amd import femasm as F, synth
dev = torch.device('cuda:0')
n = 128; M = N = P = n + 1
import numpy as np
from oracle import oracle as O
co = torch.from_numpy(O.uniform_coords(3, M, N, P)).to(dev)
plan = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=co)
props = torch.empty(n**3 * 400, dtype=torch.float64, device=dev)
plan.fill_properties(props, 90.0, 0.28, 400)
vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev); b = torch.empty(plan.nrows, dtype=torch.float64, device=dev)
def run(label, asyn, stats):
    plan.set_async(1 if asyn else 0)
    for _ in range(5): plan.assemble_system(vals, b, material=props, stride=400, bc=F.BC_S3_ALLFACES_EXACT, want_stats=stats)
    plan.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 50
    for _ in range(K): plan.assemble_system(vals, b, material=props, stride=400, bc=F.BC_S3_ALLFACES_EXACT, want_stats=stats)
    plan.sync(); torch.cuda.synchronize()
    print(label, (time.perf_counter() - t0) / K * 1e3, 'ms/step')
run('sync + stats ', False, True)
run('sync no stats', False, False)
run('async no stats', True, False)
run('sync + stats ', False, True)
