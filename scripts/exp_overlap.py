"""K assembly and b assembly of the headline workload on two streams at once against one after the other
(two plans on the same mesh; K-only call on stream 1, b-only call on stream 2)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from petsc_amd import femasm as F, synth

dev = torch.device("cuda:0")
nel = 128
M = nel + 1
coords = synth.uniform_coords(3, M, M, M, device=dev)
pk = F.Plan(F.ELASTIC3D_Q1, M, M, M, coords=coords)
pb = F.Plan(F.ELASTIC3D_Q1, M, M, M, coords=coords)
props = torch.empty(pk.nel * 400, dtype=torch.float64, device=dev)
pk.fill_properties(props, 90.0, 0.28, 400)
vals = torch.empty(pk.nnz, dtype=torch.float64, device=dev)
b = torch.empty(pk.nrows, dtype=torch.float64, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for p in (pk, pb):
    p.set_async(True)


def run(two_streams, n=20):
    pk.set_stream(s1.cuda_stream)
    pb.set_stream((s2 if two_streams else s1).cuda_stream)
    for it in range(n + 3):
        if it == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        pk.assemble_system(vals, None, material=props, stride=400, bc=F.BC_S3_ALLFACES_EXACT)
        pb.assemble_system(None, b, material=props, stride=400, bc=F.BC_S3_ALLFACES_EXACT)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for rep in range(2):
    print("one stream  : %.3f ms per K + b" % run(False))
    print("two streams : %.3f ms per K + b" % run(True))
pk.sync(); pb.sync()
