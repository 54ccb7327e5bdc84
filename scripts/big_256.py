"""One-off check at the north star's largest mesh on ONE GPU: S3 elastic 256^3 (50.9 M rows, 4.1e9 stored values =
33 GB: every value offset beyond 2^32 is exercised).  Uniform mesh (level-2 kernels) and perturbed mesh (general
kernels); a 7^3-node window in the far corner against the oracle, no unwritten slot, step time.
    python scripts/big_256.py [nel]
(the oracle is used as the checker only, as in tests/)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from petsc_amd import femasm as F, slab, synth
from oracle import oracle as O

O.build()
dev = torch.device("cuda:0")
nel = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = nel + 1


def node_start(i, j, k):
    pairs = slab._pre(k, M) * slab._tot(M) * slab._tot(M) + slab._span(k, M) * (slab._pre(j, M) * slab._tot(M) + slab._span(j, M) * slab._pre(i, M))
    return 9 * pairs


for name, perturb in (("uniform", False), ("perturbed", True)):
    coords = synth.uniform_coords(3, M, M, M, device=dev)
    if perturb:
        coords = synth.perturb_coords(coords, 3, M, M, M)
    plan = F.Plan(F.ELASTIC3D_Q1, M, M, M, coords=coords)
    props = torch.empty(plan.nel * 400, dtype=torch.float64, device=dev)
    plan.fill_properties(props, 90.0, 0.28, 400)
    vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
    b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
    st = plan.assemble_system(vals, b, material=props, stride=400, bc=F.BC_S3_ALLFACES_EXACT, want_stats=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        st = plan.assemble_system(vals, b, material=props, stride=400, bc=F.BC_S3_ALLFACES_EXACT, want_stats=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    nan = bool(torch.isnan(vals).any()) or bool(torch.isnan(b).any())
    n, (i0, j0, k0) = 7, (M - 8, M - 8, M - 8)
    c4 = coords.reshape(M, M, M, 3)[k0:k0 + n, j0:j0 + n, i0:i0 + n].contiguous().cpu().numpy().reshape(-1)
    p4 = props.reshape(nel, nel, nel, 400)[k0:k0 + n - 1, j0:j0 + n - 1, i0:i0 + n - 1].contiguous().cpu().numpy().reshape(-1)
    rps, cis = O.pattern(3, 3, n, n, n)
    Ks, bs = O.assemble_elastic3d(n, n, n, c4, p4, rps, cis)
    num = den = nb = db = 0.0
    for k in range(1, n - 1):
        for j in range(1, n - 1):
            for i in range(1, n - 1):
                g0 = node_start(i0 + i, j0 + j, k0 + k)
                g = vals[g0:g0 + 243].cpu().numpy()
                rs = 3 * (i + n * (j + n * k))
                s = Ks[rps[rs]:rps[rs + 3]]
                num += np.sum((g - s) ** 2); den += np.sum(s ** 2)
                gn = 3 * ((i0 + i) + M * ((j0 + j) + M * (k0 + k)))
                bg = b[gn:gn + 3].cpu().numpy()
                nb += np.sum((bg - bs[rs:rs + 3]) ** 2); db += np.sum(bs[rs:rs + 3] ** 2)
    print(f"{name}: nel={nel}^3 rows={plan.nrows} nnz={plan.nnz} ({plan.nnz * 8 / 1e9:.1f} GB) level={plan.fast_path()} "
          f"step={ms:.2f} ms ({plan.nel / ms * 1e3:.3e} el/s) K={st.ms_kernel_K:.2f} main={st.ms_kernel_main:.2f} rhs={st.ms_kernel_rhs:.2f} | "
          f"unwritten={nan} window relK={np.sqrt(num / den):.2e} relb={np.sqrt(nb / db):.2e} (value offsets up to {node_start(M - 2, M - 2, M - 2):.3e})",
          flush=True)
    assert not nan and np.sqrt(num / den) <= 1e-13 and np.sqrt(nb / db) <= 1e-13
    plan.close()
    del vals, b, props, coords
    torch.cuda.empty_cache()
