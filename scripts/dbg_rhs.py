"""b of the S3 runs path against the thread-per-element path (material base misaligned by 8 bytes) at sizes where
a workgroup of the persistent RHS kernel walks several tasks."""
import sys, torch
sys.path.insert(0, '.')
from petsc_amd import femasm as F, synth
dev = torch.device('cuda:0')
for n in (24, 48, 96, 128):
    M = n + 1
    co = synth.uniform_coords(3, M, M, M, device=dev)
    plan = F.Plan(F.ELASTIC3D_Q1, M, M, M, coords=co)
    buf = torch.empty(plan.nel * 400 + 1, dtype=torch.float64, device=dev)
    props = buf[:-1]; plan.fill_properties(props, 90.0, 0.28, 400)
    b1 = torch.empty(plan.nrows, dtype=torch.float64, device=dev)
    plan.assemble_system(None, b1, material=props, stride=400, bc=F.BC_S3_ALLFACES_EXACT)
    p2 = buf[1:]; p2.copy_(props.clone())
    b2 = torch.empty(plan.nrows, dtype=torch.float64, device=dev)
    plan.assemble_system(None, b2, material=p2, stride=400, bc=F.BC_S3_ALLFACES_EXACT)
    d = (b1 - b2).abs()
    bad = (d > 1e-12 * b2.abs().max()).nonzero().flatten()
    print(n, 'rel', float(torch.linalg.norm(b1 - b2) / torch.linalg.norm(b2)), 'bad rows', bad.numel(), bad[:6].tolist())
    if bad.numel():
        r = bad // 3
        print('   node k range', int((r // (M * M)).min()), int((r // (M * M)).max()), 'j', int(((r // M) % M).min()), int(((r // M) % M).max()), 'i', int((r % M).min()), int((r % M).max()))
    plan.close()
