#!/usr/bin/env python3
"""bench.py -- global assembly throughput of the femasm engine on MI355X.

A "step" is one pass of the hot path over one synthetic mesh already resident in HBM:
K (CSR values) + b + Dirichlet rows of one `femasm_assemble_system` call (plus, for N > 1,
the slab-interface exchange over RCCL and `femasm_interface_apply`).

Default workload (N = 1): BASELINE.json's target configuration, SGFEM_elastic3D Q1 hex on a
128^3 mesh (24x24 local matrices, E=90, nu=0.28, all-face Dirichlet).  For N GPUs the mesh is
128 x 128 x (128*N) elements cut into N z-slabs by the DMDA rule (weak scaling; at N = 8 it has
the element count of 256^3).  Other configurations: --workload poisson3d|elastic2d, --nel.

`python3 bench.py --gpus N` without a launcher starts its N ranks itself (torch.distributed.run as a child
process, before this process touches the GPU) and exits with the child's code.  The default one-GPU line also
carries a `general_geometry` sub-record: the same mesh with its interior nodes moved ("report both").

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="elastic3d", choices=["elastic3d", "poisson3d", "elastic2d", "enriched3d"])
    ap.add_argument("--n-enr", type=int, default=1, help="enriched3d: enrichment families per node (dof = 3(1+n_enr))")
    ap.add_argument("--enr-contraction", default="default", choices=["default", "node", "valu", "mfma"],
                    help="enriched3d: node-centric single pass (default), or element matrices contracted on the FP64 VALU / "
                         "v_mfma_f64_16x16x4 (n_enr = 1) followed by the segmented reduction")
    ap.add_argument("--nel", type=int, default=0, help="elements per direction (default: 128 / 256 / 2048)")
    ap.add_argument("--perturb", action="store_true", help="general (non-affine) geometry variant")
    ap.add_argument("--uniform-material", action="store_true", help="uniform E,nu instead of the AoS array")
    ap.add_argument("--element-soa", action="store_true",
                    help="elastic kinds, one GPU: FEMASM_MAT_ELEMENT_SOA (E and nu per element, two planes) instead of the\n"
                         "drivers' AoS records; carries no loads, so use with --k-only")
    ap.add_argument("--manufactured-loads", action="store_true",
                    help="elastic3d: FEMASM_MAT_MANUFACTURED (f and grad u evaluated in the RHS kernel, constant E, nu) "
                         "instead of streaming the drivers' AoS records")
    ap.add_argument("--strong", action="store_true", help="fixed global mesh split over N slabs")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="femasm_plan_set_option (kernel variants / tile parameters for A/B runs), repeatable")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-general", action="store_true",
                    help="skip the general_geometry sub-record (the same mesh perturbed, timed after the main variant)")
    ap.add_argument("--no-peak", action="store_true", help="skip the device copy / fill rates (roofline.measured_peak)")
    ap.add_argument("--k-only", action="store_true", help="time K assembly only (no b)")
    ap.add_argument("--fast-level", type=int, default=2, choices=[0, 1, 2],
                    help="stiffness kernel: 2 box / 1 affine / 0 general (falls back by itself when the mesh "
                         "does not qualify)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo to rehearse "
                    "several ranks on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: all ranks use GPU 0")
    ap.add_argument("--check", action="store_true", help="after the timed steps, assemble the same GLOBAL mesh as one "
                    "plan on this rank's GPU and report rel(K), rel(b) of this rank's slab rows against it (max over "
                    "ranks) in the JSON line; small meshes only")
    return ap.parse_args()


def host_cpu():
    """Model name and core count of the host the CPU baseline runs on."""
    model, ncpu = "unknown", os.cpu_count()
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        ncpu_avail = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu_avail = ncpu
    return model, ncpu, ncpu_avail


def _cfg1_worker(_):
    from oracle import oracle as O
    n = 32
    M = n + 1
    coords = O.uniform_coords(3, M, M, M)
    rp, ci = O.pattern(3, 1, M, M, M)
    t0 = time.perf_counter()
    O.assemble_poisson3d(M, M, M, coords, rp, ci)
    return time.perf_counter() - t0


def cpu_baseline(workload):
    """Serial CPU assembly (oracle = restatement of the reference's loops, 1 core) on a bounded
    sample of the same workload.  This is the ONLY place bench.py touches oracle/."""
    import numpy as np
    from oracle import oracle as O
    if workload == "elastic3d":
        n = 64
        M = n + 1
        coords = O.uniform_coords(3, M, M, M)
        props = O.props_s3(n, n, n, coords)
        rp, ci = O.pattern(3, 3, M, M, M)
        t0 = time.perf_counter()
        K, b = O.assemble_elastic3d(M, M, M, coords, props, rp, ci)
        O.bc_s3(M, M, M, coords, rp, ci, K, b)
        dt = time.perf_counter() - t0
        nel, sample = n ** 3, "64^3 sub-mesh of the 128^3 workload (K + b + BC), scaled per element"
    elif workload == "enriched3d":
        n = 10
        M = n + 1
        coords = O.uniform_coords(3, M, M, M)
        psi = [lambda x, y, z: (x * x + 0.5 * y * z + 0.25 * z, 2 * x, 0.5 * z, 0.5 * y + 0.25)]
        props = O.props_enriched(n, n, n, coords, 1, psi)
        rp, ci = O.pattern(3, 6, M, M, M)
        t0 = time.perf_counter()
        K, b = O.assemble_enriched3d(M, M, M, coords, props, rp, ci, 1)
        O.bc_enriched3d(M, M, M, coords, props, rp, ci, K, b, 1)
        dt = time.perf_counter() - t0
        nel, sample = n ** 3, ("10^3 sub-mesh of the 128^3 enriched workload, n_enr = 1 (naive 48x48 B^T D B per Gauss "
                               "point as S3:507-515 would do it; no reference code exists for this configuration)")
    elif workload == "poisson3d":
        n = 160
        M = n + 1
        coords = O.uniform_coords(3, M, M, M)
        rp, ci = O.pattern(3, 1, M, M, M)
        t0 = time.perf_counter()
        O.assemble_poisson3d(M, M, M, coords, rp, ci)
        dt = time.perf_counter() - t0
        nel, sample = n ** 3, "160^3 sub-mesh (K + b with in-loop BC)"
    else:
        n = 2560
        M = n + 1
        coords = O.uniform_coords(2, M, M)
        props = O.props_e2(n, n, coords)
        rp, ci = O.pattern(2, 2, M, M)
        t0 = time.perf_counter()
        K, b = O.assemble_elastic2d(M, M, coords, props, rp, ci)
        O.bc_e2(M, M, rp, ci, K, b)
        dt = time.perf_counter() - t0
        nel, sample = n * n, "2560^2 sub-mesh (K + b + BC)"
    out = {"value": nel / dt, "unit": "elements/s", "cores": 1, "kind": "port", "sample": sample,
           "seconds": round(dt, 3)}
    # BASELINE config 1 in full (FEM_poisson3D 32^3, serial: the reference's own CPU-runnable case, K + b with the
    # in-loop Dirichlet rows), then the same case on every core this process may use at once (independent copies,
    # one per core: a context figure for "all cores", the reference itself is serial), and the host CPU
    t1 = _cfg1_worker(0)
    model, ncpu, navail = host_cpu()
    out["config1_poisson3d_32"] = {"value": 32 ** 3 / t1, "unit": "elements/s", "cores": 1, "seconds": round(t1, 4),
                                   "sample": "32^3 in full"}
    nw = max(1, min(navail, 16))
    # Worker processes are forked HERE, before this process has imported torch or made any HIP call (main() runs
    # cpu_baseline first), and they end by themselves (close + join, never terminate): a forked copy of a process that
    # holds the GPU, killed by SIGTERM, ran a profiler's teardown in round 2 and segfaulted.  Under a profiler that
    # initialises the GPU before main() (rocprofv3 preloads its tool) no worker is forked at all.
    preload = os.environ.get("LD_PRELOAD", "") + os.environ.get("ROCP_TOOL_LIBRARIES", "")
    if "rocprof" in preload or "roctracer" in preload or "torch" in sys.modules:
        out["all_cores_context"] = {"skipped": "GPU already initialised in this process (profiler): no fork"}
    else:
        try:
            import multiprocessing as mp
            pool = mp.get_context("fork").Pool(nw)
            try:
                t0 = time.perf_counter()
                pool.map(_cfg1_worker, range(nw))
                wall = time.perf_counter() - t0
            finally:
                pool.close()
                pool.join()
            out["all_cores_context"] = {"value": nw * 32 ** 3 / wall, "unit": "elements/s", "cores": nw,
                                        "sample": f"{nw} independent copies of config 1, one per core"}
        except Exception as exc:  # a box that forbids fork: the serial figure stands alone
            out["all_cores_context"] = {"error": str(exc)}
    out["host_cpu"] = {"model": model, "logical_cpus": ncpu, "usable_cpus": navail}
    return out


def measured_traffic(wl, nel, world, args):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/*/traffic.json,
    FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc runs of this same command and corrected
    as MI355X_MICROARCH.md prescribes).  None when this exact configuration was not profiled."""
    if world != 1 or args.uniform_material or args.k_only or args.fast_level != 2 or args.element_soa:
        return None
    if wl == "enriched3d" and args.n_enr != 1:
        return None
    variant = ("general" if args.perturb else "manufactured" if args.manufactured_loads else
               args.enr_contraction if (wl == "enriched3d" and args.enr_contraction in ("mfma", "valu")) else "default")
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for rnd in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:  # the latest round that profiled it wins
        f = os.path.join(pdir, rnd, "traffic.json")
        if os.path.exists(f):
            for rec in json.load(open(f)).get("kernels", []):
                if (rec.get("workload") == wl and rec.get("nel") == nel
                        and rec.get("variant", "default") == variant):
                    best = rec.get("hbm_bytes_per_launch")
    return best


def launch_ranks(args):
    """`python3 bench.py --gpus N` without a launcher: start the N ranks as a child torch.distributed.run (this
    process has made no GPU call and makes none), pass its output through, exit with its code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def workload_traits(F, args):
    wl = args.workload
    if wl == "elastic3d":
        t = dict(kind=F.ELASTIC3D_Q1, nsd=3, dof=3, ngp=8, bc=F.BC_S3_ALLFACES_EXACT, stride=400, nel=args.nel or 128,
                 ref="SGFEM/SGFEM_elastic3D.cpp")
    elif wl == "enriched3d":
        t = dict(kind=F.ELASTIC3D_ENRICHED, nsd=3, dof=3 * (1 + args.n_enr), ngp=8, bc=F.BC_S3_ALLFACES_EXACT,
                 stride=408 + 64 * args.n_enr, nel=args.nel or 128,
                 ref="SGFEM_real/SGFEM_elastic3D.cpp (enriched; specification of this engine)")
    elif wl == "poisson3d":
        t = dict(kind=F.POISSON3D_Q1, nsd=3, dof=1, ngp=8, bc=F.BC_POISSON_INLOOP, stride=0, nel=args.nel or 256,
                 ref="FEM_poisson3D/FEM.cpp")
    else:
        t = dict(kind=F.PLANE_STRAIN2D_Q1, nsd=2, dof=2, ngp=4, bc=F.BC_E2_COMPRESSION, stride=24, nel=args.nel or 2048,
                 ref="FEM_elsatic2D/FEM.cpp")
    return t


# FP64 operations per element of the GENERAL-geometry stiffness kernels, counted on the reference's algorithm with the
# structural zeros of B and D skipped and nothing shared between Gauss points: per Gauss point the
# Jacobian (NSD*NSD*NEN FMAs), its inverse and determinant, the physical gradients (NSD*NSD*NEN), the scaled
# coefficients, and one dof x dof block per node pair (P3: 3 FMAs; E2: 10; S3: 21 FMAs + the 9 products D*grad per node).
GENERAL_FLOPS_PER_ELEMENT = {
    "poisson3d": 8 * (2 * 72 + 60 + 2 * 72 + 64 * 2 * 3 + 8 * 3),
    "elastic2d": 4 * (2 * 16 + 12 + 2 * 16 + 12 + 16 * 2 * 10 + 4 * 6),
    "elastic3d": 8 * (2 * 72 + 60 + 2 * 72 + 20 + 64 * 2 * 21 + 8 * 9),
}

# k_assemble_s3_hex2 (level 1): per Gauss point the same geometry (2*72 + 60 + 2*72 + the 96 products of the reference
# gradients' factors), 64 node pairs x 9 FMAs on w detJ-scaled gradients (+ 24 products), and once per element 64 x 21.
HEX_FLOPS_PER_ELEMENT = 8 * (2 * 72 + 60 + 2 * 72 + 96 + 64 * 2 * 9 + 24) + 64 * 21


def kernel_name(wl, level, args):
    if wl == "enriched3d":
        mode = args.enr_contraction if args.enr_contraction != "default" else "node"
        return ((f"k_enr_box<S={1 + args.n_enr}>" if level == 2 else
                 f"k_enr_node<S={1 + args.n_enr},{'one material per element' if level == 1 else 'general'}>")
                if mode == "node" else
                f"k_enr_elem<S={1 + args.n_enr},{mode}> + k_enr_gather (all of K)")
    if wl == "elastic2d":
        return "k_assemble_e2_rect" if level == 2 else "k_assemble_ke_e2"
    if wl == "poisson3d":
        return {2: "k_assemble_p3_rect", 1: "k_assemble_rowsweep<POISSON3D>"}.get(level, "k_assemble_p3_hex4")
    return {2: "k_assemble_s3_unit", 1: "k_assemble_s3_hex2"}.get(level, "k_assemble_sweep<ELASTIC3D>")


def measure_peak(torch, nbytes, dev):
    """Device fill and copy rates in this process, outside every timed region ("fraction against both
    nominal and measured"): torch's fill of `nbytes` and a device-to-device copy of `nbytes` (read + write counted)."""
    n = max(int(nbytes) // 8, 1 << 20)
    a = torch.empty(n, dtype=torch.float64, device=dev)
    bb = torch.empty(n, dtype=torch.float64, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for _ in range(2):
        a.zero_()
        bb.copy_(a)
    torch.cuda.synchronize()
    reps = 5
    ev[0].record()
    for _ in range(reps):
        a.zero_()
    ev[1].record()
    ev[2].record()
    for _ in range(reps):
        bb.copy_(a)
    ev[3].record()
    torch.cuda.synchronize()
    fill = 8.0 * n * reps / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9
    copy = 16.0 * n * reps / (ev[2].elapsed_time(ev[3]) * 1e-3) / 1e9
    del a, bb
    return {"fill_GBps": fill, "copy_GBps": copy, "bytes": 8 * n,
            "how": "torch fill_ / copy_ of this many bytes on the bench stream, HIP events, 5 repeats"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # the CPU baseline first: its worker pool is forked before this process imports torch or touches the GPU
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.workload)

    import torch
    import torch.distributed as dist
    from petsc_amd import femasm as F
    from petsc_amd import slab, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the femasm engine has no CPU path")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl_ranks = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        rccl_ranks = dist.get_world_size()

    wl = args.workload
    tr = workload_traits(F, args)
    kind, nsd, dof, ngp, bc, nel, ref = tr["kind"], tr["nsd"], tr["dof"], tr["ngp"], tr["bc"], tr["nel"], tr["ref"]

    # global mesh: weak scaling stacks N copies of the single-GPU mesh along the slab direction
    nslow = nel if args.strong else nel * world
    M = N = nel + 1
    if nsd == 3:
        P, SL = nslow + 1, nslow + 1
    else:
        N, P, SL = nslow + 1, 1, nslow + 1
    ks, ke = slab.slab_ranges(SL, world)[rank]
    cks, cke = max(ks - 1, 0), min(ke + 1, SL)
    els, ele = max(ks - 1, 0), ke - 1
    E, nu = (90.0, 0.28) if wl in ("elastic3d", "enriched3d") else (1.0, 0.33)

    vals = b = None  # output buffers, shared by the variants measured in this process

    def run_variant(perturb):
        """Plan + inputs for one geometry variant, W warm-up steps, K timed steps: (seconds, stats, plan facts)."""
        nonlocal vals, b
        stride = tr["stride"]
        coords = synth.uniform_coords(nsd, M, N, P, cks, cke, device=dev)
        if perturb:
            coords = synth.perturb_coords(coords, nsd, M, N, P, k0=cks)
        material = None
        plan = F.Plan(kind, M, N, P, coords=coords, ks=ks, ke=ke, cks=cks, cke=cke, device=local_rank,
                      n_enr=args.n_enr if wl == "enriched3d" else 0)
        plan.set_stream(torch.cuda.current_stream().cuda_stream)
        plan.set_fast_path(args.fast_level)
        for kv in args.option:
            name, _, value = kv.partition("=")
            plan.set_option(name, int(value or 1))
        if wl == "enriched3d":
            if args.enr_contraction != "default":
                plan.set_enriched_contraction({"valu": 0, "mfma": 1, "node": 2}[args.enr_contraction])
            material = synth.props_enriched(M, N, P, coords, n_enr=args.n_enr, e0=els, e1=ele, ck0=cks, E=E, nu=nu)
        elif wl != "poisson3d" and not args.uniform_material and not args.manufactured_loads:
            # the drivers' GaussPointCoefficients fill (S3:1243-1363 / E2:618-670), on device (row N1)
            material = torch.empty(plan.nel * stride, dtype=torch.float64, device=dev)
            plan.fill_properties(material, E, nu, stride)
            if args.element_soa:
                assert world == 1, "--element-soa: one GPU"
                material = torch.cat([torch.full((plan.nel,), E, dtype=torch.float64, device=dev),
                                      torch.full((plan.nel,), nu, dtype=torch.float64, device=dev)])
                stride = plan.nel
        if vals is None:
            vals = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
            b = None if args.k_only else torch.empty(plan.nrows, dtype=torch.float64, device=dev)

        send = recv = None
        if world > 1:
            (_, sc), (_, rc) = plan.interface_buffers()
            send = torch.empty(max(sc, 1), dtype=torch.float64, device=dev)
            recv = torch.empty(max(rc, 1), dtype=torch.float64, device=dev)
            plan.interface_set_buffers(send if sc else None, recv if rc else None)

        k_ms, main_ms, rhs_ms = [], [], []

        def step(collect):
            reqs = None
            if world > 1 and wl == "enriched3d":
                # (no split call for the enriched kind: the send buffer is filled by the assembly itself)
                st = plan.assemble_system(vals, b, material=material, stride=stride, bc=bc, e_ks=els, want_stats=True)
                if args.backend != "nccl":
                    plan.sync()
                slab.exchange_interface(send if rank > 0 else None, recv if rank < world - 1 else None, rank, world)
                plan.interface_apply(vals, b, bc=bc)
            else:
                if world > 1:
                    # the rows owed to the rank below first, so that their transfer overlaps the bulk of the assembly
                    plan.assemble_interface(material=material, stride=stride, bc=bc, E=E, nu=nu, e_ks=els)
                    if args.backend != "nccl":
                        plan.sync()  # gloo reads the send tensor from the host side: wait for the kernels
                    reqs = slab.exchange_interface_start(send if rank > 0 else None,
                                                         recv if rank < world - 1 else None, rank, world)
                st = plan.assemble_system(vals, b, material=material, stride=stride, bc=bc, E=E, nu=nu, e_ks=els,
                                          want_stats=True,
                                          manufactured=args.manufactured_loads and wl == "elastic3d",
                                          element_soa=args.element_soa and material is not None)
                if world > 1:
                    slab.exchange_interface_finish(reqs)
                    plan.interface_apply(vals, b, bc=bc)
            if collect:
                k_ms.append(st.ms_kernel_K)
                main_ms.append(st.ms_kernel_main)
                rhs_ms.append(st.ms_kernel_rhs)
            return st

        for _ in range(args.warmup):
            st = step(False)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st = step(True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        facts = {"level": plan.fast_path(), "k_ms": sum(k_ms) / len(k_ms), "main_ms": sum(main_ms) / len(main_ms),
                 "rhs_ms": sum(rhs_ms) / len(rhs_ms), "stride": stride, "has_material": material is not None}
        return dt, st, plan, facts

    def roofline_of(st, facts, perturb, peak):
        level = facts["level"]
        if facts["main_ms"] > 0.0 and st.bytes_main > 0.0:
            achieved = st.bytes_main / (facts["main_ms"] * 1e-3) / 1e9
        elif facts["k_ms"] > 0.0:  # no interior kernel was launched (a mesh or slab without interior nodes): all of K
            achieved = st.bytes_K / (facts["k_ms"] * 1e-3) / 1e9
        else:
            achieved = None
        pargs = argparse.Namespace(**vars(args))
        pargs.perturb = perturb
        roof = {
            "kernel": kernel_name(wl, level, args),
            "kernel_level": level,
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": None if achieved is None else achieved / HBM_PEAK_GBS,
            "traffic": measured_traffic(wl, nel, world, pargs),
            "algorithmic_bytes_per_launch": st.bytes_main,
            "avg_launch_ms": facts["main_ms"],
        }
        if peak is not None and achieved is not None:
            # the dominant kernels are store streams: the fill rate is their measured ceiling; the copy rate is the
            # guide's "achievable HBM" figure (6.29 TB/s on its box)
            roof["measured_peak"] = peak
            roof["frac_of_measured_fill"] = achieved / peak["fill_GBps"]
            roof["frac_of_measured_copy"] = achieved / peak["copy_GBps"]
        general = level == 0 or (level == 1 and wl == "elastic3d")  # the kernels that do the per-Gauss-point geometry
        if general and wl in GENERAL_FLOPS_PER_ELEMENT and facts["main_ms"] > 0.0:
            fl = GENERAL_FLOPS_PER_ELEMENT[wl] * float(st.nel)
            roof["fp64_tflops"] = fl / (facts["main_ms"] * 1e-3) / 1e12
            roof["fp64_peak_tflops"] = 78.6
            roof["fp64_flops_per_element"] = GENERAL_FLOPS_PER_ELEMENT[wl]
            if level == 1:
                # k_assemble_s3_hex2 takes the material out of the Gauss-point loop: what it executes per element
                roof["fp64_flops_executed_per_element"] = HEX_FLOPS_PER_ELEMENT
                roof["fp64_tflops_executed"] = HEX_FLOPS_PER_ELEMENT * float(st.nel) / (facts["main_ms"] * 1e-3) / 1e12
        return roof

    dt, st, plan, facts = run_variant(args.perturb)
    stride = facts["stride"]

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    nel_local = torch.tensor([float(st.nel)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(nel_local, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    nel_total = float(nel_local.item())

    check = None
    if args.check:
        # the same GLOBAL mesh as one plan on this rank's GPU; this rank's slab rows against the global system's rows
        gco = synth.uniform_coords(nsd, M, N, P, device=dev)
        if args.perturb:
            gco = synth.perturb_coords(gco, nsd, M, N, P)
        gplan = F.Plan(kind, M, N, P, coords=gco, device=local_rank, n_enr=args.n_enr if wl == "enriched3d" else 0)
        gmat = None
        if facts["has_material"]:
            if wl == "enriched3d":
                gmat = synth.props_enriched(M, N, P, gco, n_enr=args.n_enr, E=E, nu=nu)
            else:
                gmat = torch.empty(gplan.nel * stride, dtype=torch.float64, device=dev)
                gplan.fill_properties(gmat, E, nu, stride)
        gv = torch.empty(gplan.nnz, dtype=torch.float64, device=dev)
        gb = torch.empty(gplan.nrows, dtype=torch.float64, device=dev)
        gplan.assemble_system(gv, gb, material=gmat, stride=stride, bc=bc, E=E, nu=nu)
        lo = slab.layer_nnz(nsd, dof, M, N, P, 0, ks)
        plane = (M * N if nsd == 3 else M) * dof
        relK = float(torch.linalg.norm(vals - gv[lo:lo + plan.nnz]) / torch.linalg.norm(gv[lo:lo + plan.nnz]))
        relb = 0.0
        if b is not None:
            gbs = gb[plane * ks:plane * ke]
            relb = float(torch.linalg.norm(b - gbs) / torch.linalg.norm(gbs))
        cmax = torch.tensor([relK, relb], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
        check = {"rel_K": float(cmax[0].item()), "rel_b": float(cmax[1].item()),
                 "against": "one-plan assembly of the global mesh, this rank's rows, max over ranks"}
        gplan.close()
        del gv, gb, gmat, gco

    peak = None
    if rank == 0 and world == 1 and not args.no_peak:
        peak = measure_peak(torch, min(8 * plan.nnz, 4 << 30), dev)

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        k_avg_ms = facts["k_ms"]
        mesh_txt = ("x".join(str(v - 1) for v in ((M, N, P) if nsd == 3 else (M, N))) + " elements")
        if world > 1:
            scale_txt = (f"{mesh_txt} fixed, cut into {world} z-slabs (strong scaling)" if args.strong else
                         f"{nel}^{nsd} elements per GPU-slab, {world} slabs stacked along z = {mesh_txt} "
                         f"(weak scaling; 8 slabs of 128^3 hold the element count of 256^3)")
        else:
            scale_txt = f"{mesh_txt} on 1 GPU"
        out = {
            "metric": "global assembly elements/sec + achieved HBM GB/s fraction",
            "value": nel_total / (dt / args.steps),
            "unit": "elements/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{ref} Q1 {wl}, {scale_txt}"
                            f"{', general geometry' if args.perturb else ', uniform DMDA coordinates'}, "
                            f"K + b + Dirichlet rows{' (K only)' if args.k_only else ''}",
                "global_elements": int(nel_total),
                "global_mesh": (mesh_txt + ", " + ("fixed (strong scaling)" if args.strong else
                                f"{nel} per direction per GPU stacked along the slab direction (weak scaling)")),
                "rows_per_gpu": int(st.nrows),
                "nnz_per_gpu": int(st.nnz),
                "material": ("manufactured loads evaluated in-kernel, constant E, nu" if args.manufactured_loads else
                             "uniform") if not facts["has_material"] else
                            ("per-element (E, nu) planes (FEMASM_MAT_ELEMENT_SOA)" if args.element_soa else
                             f"AoS GaussPointCoefficients stride {stride}"),
                "parallelism": (f"z-slab x{world}, one process per GPU, interface rows exchanged over "
                                f"{'RCCL' if args.backend == 'nccl' else args.backend} p2p" if world > 1 else "1 GPU"),
                "rccl_ranks": rccl_ranks,
            },
            "roofline": roofline_of(st, facts, args.perturb, peak),
            "k_ms": k_avg_ms,
            "rhs_ms": facts["rhs_ms"],
        }
        # the whole K assembly of this rank (every stiffness kernel incl. Dirichlet rows and the pre-pass) against
        # SURVEY 8(d)'s formula (femasm_stats.bytes_K = 8 nnz + 8 NSD nnodes + 16 NGP nel [E,nu per Gauss point])
        k_gbs = st.bytes_K / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0.0 else None
        out["k_assembly"] = {"algorithmic_bytes": st.bytes_K, "ms": k_avg_ms, "achieved_GBps": k_gbs,
                             "frac": None if k_gbs is None else k_gbs / HBM_PEAK_GBS}
        if check is not None:
            out["check"] = check
    plan.close()

    # "a second 'general geometry' variant ... report both" -- the same mesh with its interior nodes moved
    # by 0.2 h (no element is a parallelepiped), same material, same steps, timed after the main variant
    if (rank == 0 and world == 1 and not args.perturb and not args.no_general and wl != "enriched3d"
            and args.fast_level == 2):
        del plan
        gdt, gst, gplan, gfacts = run_variant(True)
        groof = roofline_of(gst, gfacts, True, peak)
        gk = gfacts["k_ms"]
        out["general_geometry"] = {
            "workload": f"the same {nel}^{nsd} mesh, interior nodes moved by 0.2 h sin(..) (synth.perturb_coords)",
            "value": float(gst.nel) / (gdt / args.steps),
            "unit": "elements/s",
            "ms_per_step": 1e3 * gdt / args.steps,
            "k_ms": gk,
            "rhs_ms": gfacts["rhs_ms"],
            "roofline": groof,
            "k_assembly": {"algorithmic_bytes": gst.bytes_K, "ms": gk,
                           "frac": gst.bytes_K / (gk * 1e-3) / 1e9 / HBM_PEAK_GBS if gk > 0.0 else None},
        }
        gplan.close()
    if rank == 0:
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
