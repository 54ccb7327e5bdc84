"""world_size-2 (and 3) CPU test of the slab path over gloo: the partition, the interface buffer
shapes and the point-to-point exchange of petsc_amd/slab.py, with the oracle standing in for the
per-slab compute (the GPU kernels are covered by the -m gpu tests, incl. a multi-slab one)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, kind, dims, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from petsc_amd import slab
        M, N, P = dims
        nsd, dof = (3, 1) if kind == "p3" else (3, 3)
        coords = O.perturb_coords(O.uniform_coords(3, M, N, P), 3, M, N, P)
        rp, ci = O.pattern(nsd, dof, M, N, P)
        ks, ke = slab.slab_ranges(P, world)[rank]
        e0, e1 = slab.element_range(ks, ke)
        plane = M * N * dof
        if kind == "p3":
            K, b = O.assemble_poisson3d(M, N, P, coords, rp, ci, e0=e0, e1=e1)
            flags = np.zeros(plane * P, dtype=np.uint8)
        else:
            props = O.props_s3(M - 1, N - 1, P - 1, coords)
            K, b = O.assemble_elastic3d(M, N, P, coords, props, rp, ci, e0=e0, e1=e1)
            flags = O.bc_flags("s3", M, N, P)
        # what this rank owes to the rank below: rows of node layer ks-1, CSR layout, then the b slice
        sc, rc = slab.interface_counts(nsd, dof, M, N, P, ks, ke)
        send = recv = None
        if ks > 0:
            r0, r1 = plane * (ks - 1), plane * ks
            full = K[rp[r0]:rp[r1]]
            packed = slab.pack_layer_rows(nsd, dof, M, N, P, ks - 1, full)
            chk = np.zeros_like(full)      # nothing but zeros is left behind: the packed form loses no value
            slab.add_packed_rows(nsd, dof, M, N, P, ks - 1, chk, packed)
            assert np.array_equal(chk, full)
            send = torch.from_numpy(np.concatenate([packed, b[r0:r1]]))
            assert send.numel() == sc
        if ke < P:
            recv = torch.empty(rc, dtype=torch.float64)
        if kind == "s3":  # the non-blocking pair bench.py uses to overlap the transfer with the assembly
            slab.exchange_interface_finish(slab.exchange_interface_start(send, recv, rank, world))
        else:
            slab.exchange_interface(send, recv, rank, world)
        # own + received on the top owned layer, Dirichlet rows untouched (elastic: rows are set afterwards)
        lo, hi = plane * ks, plane * ke
        Kl, bl = K[rp[lo]:rp[hi]].copy(), b[lo:hi].copy()
        if recv is not None:
            t0, t1 = plane * (ke - 1), plane * ke
            nk = rc - plane
            top = Kl[rp[t0] - rp[lo]:]     # (a view: added to in place)
            slab.add_packed_rows(nsd, dof, M, N, P, ke - 1, top, recv[:nk].numpy())
            bl[t0 - lo:] += recv[nk:].numpy()
        q.put((rank, ks, ke, Kl, bl))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,kind,dims", [(2, "p3", (4, 3, 7)), (2, "s3", (3, 3, 5)), (3, "p3", (3, 4, 8))])
def test_slab_exchange_reproduces_the_serial_system(O, world, kind, dims):
    port = 29500 + (os.getpid() % 2000) + world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, dims, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    M, N, P = dims
    nsd, dof = (3, 1) if kind == "p3" else (3, 3)
    coords = O.perturb_coords(O.uniform_coords(3, M, N, P), 3, M, N, P)
    rp, ci = O.pattern(nsd, dof, M, N, P)
    if kind == "p3":
        K, b = O.assemble_poisson3d(M, N, P, coords, rp, ci)
    else:
        K, b = O.assemble_elastic3d(M, N, P, coords, O.props_s3(M - 1, N - 1, P - 1, coords), rp, ci)
    Kc = np.concatenate([p[3] for p in parts])
    bc = np.concatenate([p[4] for p in parts])
    assert Kc.shape == K.shape
    assert np.linalg.norm(Kc - K) <= 1e-14 * np.linalg.norm(K)
    assert np.linalg.norm(bc - b) <= 1e-14 * np.linalg.norm(b)
