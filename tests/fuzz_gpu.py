"""Randomised parity run (not collected by pytest; run by name on the GPU box):

    python tests/fuzz_gpu.py --seconds 120 --seed 1

Random problem kind, mesh extents around the kernels' tile sizes, mesh type (uniform / graded rectilinear / sheared /
perturbed), material (constant / one (E,nu) per element / per Gauss point), slab count, the split interface call, the
asynchronous mode, and (--kinds enr) the enriched kind with random families, flags, record width and assembly form;
every system is compared with the CPU oracle at the tolerance of test_gpu_parity.py.  Prints one line per case and
stops at the first mismatch with the parameters that reproduce it."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TOL = 1e-13
XS = [2, 3, 4, 5, 9, 16, 17, 21, 22, 23, 31, 32, 33, 34, 43, 63, 64, 65, 66, 67, 85, 97, 127, 128, 129, 130, 131]
YS = [2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 31, 32, 33, 34, 35, 65, 66]
BUDGET = {"p3": 60000, "e2": 60000, "s3": 9000, "enr": 1500}  # elements per case: keeps the serial oracle to a few seconds


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def graded_axis(rng, n):
    h = rng.uniform(0.5, 1.5, size=n - 1)
    x = np.concatenate([[0.0], np.cumsum(h)])
    return x / x[-1]


def make_coords(O, rng, nsd, M, N, P, mesh):
    if mesh == "uniform":
        return O.uniform_coords(nsd, M, N, P) if nsd == 3 else O.uniform_coords(2, M, N)
    if mesh == "perturbed":
        c = O.uniform_coords(nsd, M, N, P) if nsd == 3 else O.uniform_coords(2, M, N)
        return O.perturb_coords(c, nsd, M, N, P) if nsd == 3 else O.perturb_coords(c, 2, M, N)
    ax = [graded_axis(rng, M), graded_axis(rng, N)] + ([graded_axis(rng, P)] if nsd == 3 else [])
    if nsd == 3:
        z, y, x = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
        c = np.stack([x, y, z], axis=-1)
    else:
        y, x = np.meshgrid(ax[1], ax[0], indexing="ij")
        c = np.stack([x, y], axis=-1)
    if mesh == "sheared":
        A = np.eye(nsd) + rng.uniform(-0.2, 0.2, size=(nsd, nsd))
        c = c @ A.T
    return np.ascontiguousarray(c.reshape(-1))


def make_case(O, F, rng, kind, dims, mesh, material):
    if kind == "p3":
        M, N, P = dims
        coords = make_coords(O, rng, 3, M, N, P, mesh)
        rp, ci = O.pattern(3, 1, M, N, P)
        K, b = O.assemble_poisson3d(M, N, P, coords, rp, ci)
        return dict(problem=F.POISSON3D_Q1, dims=(M, N, P), coords=coords, props=None, stride=0,
                    bc=F.BC_POISSON_INLOOP, rp=rp, ci=ci, K=K, b=b, nsd=3)
    if kind == "e2":
        M, N = dims
        coords = make_coords(O, rng, 2, M, N, 1, mesh)
        rp, ci = O.pattern(2, 2, M, N)
        props = O.props_e2(M - 1, N - 1, coords).reshape(-1, 24)
        ne = props.shape[0]
        props[:, 16:24] = rng.uniform(-1, 1, size=(ne, 8))
        if material == "element":
            props[:, 8:12] = rng.uniform(0.5, 2.0, size=(ne, 1))
            props[:, 12:16] = rng.uniform(0.1, 0.4, size=(ne, 1))
        elif material == "gp":
            props[:, 8:12] = rng.uniform(0.5, 2.0, size=(ne, 4))
            props[:, 12:16] = rng.uniform(0.1, 0.4, size=(ne, 4))
        props = props.reshape(-1)
        K, b = O.assemble_elastic2d(M, N, coords, props, rp, ci)
        O.bc_e2(M, N, rp, ci, K, b)
        return dict(problem=F.PLANE_STRAIN2D_Q1, dims=(M, N, 1), coords=coords, props=props, stride=24,
                    bc=F.BC_E2_COMPRESSION, rp=rp, ci=ci, K=K, b=b, nsd=2)
    if kind == "enr":
        # the enriched kind (specification of this engine) against its CPU restatement
        M, N, P = dims
        n_enr, sgfem, base = int(rng.integers(0, 4)), int(rng.integers(0, 2)), int(rng.choice([408, 416]))
        dirichlet = bool(rng.integers(0, 2))
        coords = make_coords(O, rng, 3, M, N, P, mesh)
        psi = [lambda x, y, z: (x * x + 0.5 * y * z + 0.25 * z, 2 * x, 0.5 * z, 0.5 * y + 0.25),
               lambda x, y, z: (np.sin(2 * x) * y, 2 * np.cos(2 * x) * y, np.sin(2 * x), 0 * x),
               lambda x, y, z: (z * z * x + y, z * z, 1 + 0 * x, 2 * z * x)][:n_enr]
        et = ne = None
        if n_enr and rng.integers(0, 2):
            et = (rng.uniform(size=(P, N, M)) < 0.6).astype(np.int64) * rng.integers(1, 4, size=(P, N, M))
            ne = rng.integers(0, n_enr + 1, size=(P, N, M)).astype(np.int64)
        props = O.props_enriched(M - 1, N - 1, P - 1, coords, n_enr, psi, etype=et, nenrich=ne, base=base)
        stride = O.enr_stride(n_enr, base)
        if material != "constant":
            pr = props.reshape(-1, stride)
            w = 1 if material == "element" else 8
            pr[:, 96:104] = rng.uniform(50, 150, size=(pr.shape[0], w))
            pr[:, 104:112] = rng.uniform(0.1, 0.4, size=(pr.shape[0], w))
        D = 3 * (1 + n_enr)
        rp, ci = O.pattern(3, D, M, N, P)
        K, b = O.assemble_enriched3d(M, N, P, coords, props, rp, ci, n_enr, sgfem=sgfem, base=base)
        O.bc_enriched3d(M, N, P, coords, props, rp, ci, K, b, n_enr, dirichlet=dirichlet, base=base)
        mode = int(rng.choice([0, 2, 2] + ([1] if n_enr == 1 else [])))
        return dict(problem=F.ELASTIC3D_ENRICHED, dims=(M, N, P), coords=coords, props=props, stride=stride,
                    bc=F.BC_S3_ALLFACES_EXACT if dirichlet else F.BC_NONE, rp=rp, ci=ci, K=K, b=b, nsd=3,
                    n_enr=n_enr, enr_mode=0 if sgfem else 1, contraction=mode)
    M, N, P = dims
    coords = make_coords(O, rng, 3, M, N, P, mesh)
    rp, ci = O.pattern(3, 3, M, N, P)
    props = O.props_s3(M - 1, N - 1, P - 1, coords).reshape(-1, 400)
    ne = props.shape[0]
    props[:, 184:400] = rng.uniform(-1, 1, size=(ne, 216))
    if material == "element":
        props[:, 96:104] = rng.uniform(50, 150, size=(ne, 1))
        props[:, 104:112] = rng.uniform(0.1, 0.4, size=(ne, 1))
    elif material == "gp":
        props[:, 96:104] = rng.uniform(50, 150, size=(ne, 8))
        props[:, 104:112] = rng.uniform(0.1, 0.4, size=(ne, 8))
    props = props.reshape(-1)
    K, b = O.assemble_elastic3d(M, N, P, coords, props, rp, ci)
    O.bc_s3(M, N, P, coords, rp, ci, K, b)
    return dict(problem=F.ELASTIC3D_Q1, dims=(M, N, P), coords=coords, props=props, stride=400,
                bc=F.BC_S3_ALLFACES_EXACT, rp=rp, ci=ci, K=K, b=b, nsd=3)


def run_slabs(torch, F, slab, dev, case, nslab, split, use_async=False):
    M, N, P = case["dims"]
    nsd = case["nsd"]
    SL = P if nsd == 3 else N
    plane_e = ((M - 1) * (N - 1) if nsd == 3 else (M - 1))
    ranges = slab.slab_ranges(SL, nslab)
    plans, outs, bufs, levels = [], [], [], []
    for (ks, ke) in ranges:
        cks, cke = max(ks - 1, 0), min(ke + 1, SL)
        els, ele = slab.element_range(ks, ke)
        co = torch.from_numpy(case["coords"].reshape(SL, -1)[cks:cke].reshape(-1).copy()).to(dev)
        plan = F.Plan(case["problem"], M, N, P, coords=co, ks=ks, ke=ke, cks=cks, cke=cke, n_enr=case.get("n_enr", 0))
        enr = case["problem"] == F.ELASTIC3D_ENRICHED
        if enr:
            plan.set_enriched_contraction(case["contraction"])
        elif use_async:
            plan.set_async(1)
        kw = dict(enr_mode=case["enr_mode"]) if enr else {}
        mat = None
        if case["props"] is not None:
            mat = torch.from_numpy(case["props"].reshape(-1, plane_e * case["stride"])[els:ele].reshape(-1).copy()).to(dev)
        vals = torch.full((plan.nnz,), float("nan"), dtype=torch.float64, device=dev)
        b = torch.full((plan.nrows,), float("nan"), dtype=torch.float64, device=dev)
        (_, sc), (_, rc) = plan.interface_buffers()
        send = torch.full((max(sc, 1),), float("nan"), dtype=torch.float64, device=dev)
        recv = torch.full((max(rc, 1),), float("nan"), dtype=torch.float64, device=dev)
        if nslab > 1:
            plan.interface_set_buffers(send if sc else None, recv if rc else None)
        split_here = split and nslab > 1 and not enr   # (the enriched kind has no split call)
        if split_here:
            plan.assemble_interface(material=mat, stride=case["stride"], bc=case["bc"], e_ks=els)
            if use_async:
                plan.sync()
            sent = send.clone()
        plan.assemble_system(vals, b, material=mat, stride=case["stride"], bc=case["bc"], e_ks=els, **kw)
        if use_async and not enr:
            plan.sync()
        if split_here and sc:
            assert torch.equal(sent[:sc], send[:sc]), "the full call changed the send buffer"
        levels.append(plan.fast_path())
        plans.append(plan); outs.append((vals, b)); bufs.append((send, recv, sc, rc))
    for r in range(nslab - 1):
        assert bufs[r][3] == bufs[r + 1][2]
        bufs[r][1][:bufs[r][3]].copy_(bufs[r + 1][0][:bufs[r + 1][2]])
    if nslab > 1:
        for r in range(nslab):
            plans[r].interface_apply(outs[r][0], outs[r][1], bc=case["bc"])
            if use_async and case["problem"] != F.ELASTIC3D_ENRICHED:
                plans[r].sync()
    K = np.concatenate([o[0].cpu().numpy() for o in outs])
    b = np.concatenate([o[1].cpu().numpy() for o in outs])
    for p in plans:
        p.close()
    return K, b, levels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-cases", type=int, default=100000)
    ap.add_argument("--kinds", default="p3,e2,s3")
    ap.add_argument("--split-only", action="store_true", help="only multi-slab cases with the split interface call")
    args = ap.parse_args()
    import torch
    from oracle import oracle as O
    from petsc_amd import femasm as F
    from petsc_amd import slab
    O.build()
    F.load()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    n = 0
    worst = 0.0
    while time.time() < t_end and n < args.max_cases:
        kind = rng.choice(args.kinds.split(","))
        while True:
            if kind == "e2":
                dims = (int(rng.choice(XS + [257, 513])), int(rng.choice(YS + [97, 130])))
                nel = (dims[0] - 1) * (dims[1] - 1)
            elif kind == "enr":
                dims = (int(rng.choice([2, 3, 4, 5, 9, 10, 17])), int(rng.choice([2, 3, 4, 5, 8, 9])), int(rng.choice([2, 3, 4, 5, 7, 9, 12])))
                nel = (dims[0] - 1) * (dims[1] - 1) * (dims[2] - 1)
            else:
                dims = (int(rng.choice(XS)), int(rng.choice(YS)), int(rng.choice(YS + [70])))
                nel = (dims[0] - 1) * (dims[1] - 1) * (dims[2] - 1)
            if nel <= BUDGET[kind]:
                break
        mesh = str(rng.choice(["uniform", "graded", "graded", "sheared", "perturbed"]))
        material = "none" if kind == "p3" else str(rng.choice(["constant", "element", "element", "gp"]))
        if kind == "enr" and mesh == "sheared":
            mesh = "perturbed"
        use_async = bool(rng.integers(0, 2))
        SL = dims[-1]
        nslab = int(rng.integers(1, min(4, SL // 2) + 1)) if SL >= 4 else 1
        split = bool(rng.integers(0, 2))
        if args.split_only:
            if SL < 4:
                continue
            nslab, split = max(nslab, 2), True
        tag = f"seed={args.seed} case={n} {kind} dims={dims} mesh={mesh} material={material} nslab={nslab} split={split} async={use_async}"
        case = make_case(O, F, rng, kind, dims, mesh, material)
        if kind == "enr":
            tag += f" n_enr={case['n_enr']} enr_mode={case['enr_mode']} contraction={case['contraction']} stride={case['stride']} bc={case['bc']}"
        K, b, levels = run_slabs(torch, F, slab, dev, case, nslab, split, use_async)
        eK, eb = rel(K, case["K"]), rel(b, case["b"])
        ok = (not np.isnan(K).any()) and (not np.isnan(b).any()) and eK <= TOL and eb <= TOL
        worst = max(worst, eK, eb)
        print(f"{'ok  ' if ok else 'FAIL'} {tag} levels={levels} relK={eK:.2e} relb={eb:.2e}", flush=True)
        if not ok:
            sys.exit(1)
        n += 1
    print(f"fuzz: {n} cases, worst relative error {worst:.2e}")


if __name__ == "__main__":
    main()
