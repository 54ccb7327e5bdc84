"""Host logic of the multi-GPU slab partition; no compute here.

One process per GPU; GPU r owns the node layers the DMDA rule gives rank r of a (1,1,p) process
grid, assembles the elements whose UPPER node layer it owns (the reference's
DMDAGetElementsCorners/Sizes ownership), and owes the partial rows of node layer ks-1 to rank
r-1: one point-to-point send per interface (RCCL on GPUs, gloo in the CPU tests) -- the data the
reference moves inside MatAssemblyEnd's stash and DMLocalToGlobal(ADD_VALUES).
"""
import torch
import torch.distributed as dist


def slab_ranges(nlayers, nranks):
    """PETSc DMDA split of node layers: rank r gets P/p + (P%p > r)."""
    out, s = [], 0
    for r in range(nranks):
        n = nlayers // nranks + (1 if nlayers % nranks > r else 0)
        out.append((s, s + n))
        s += n
    return out


def element_range(ks, ke):
    """Element layers assembled by the owner of node layers [ks,ke)."""
    return max(ks - 1, 0), ke - 1


def _span(i, M):
    return 1 if M == 1 else 3 - (i == 0) - (i == M - 1)


def _tot(M):
    return 1 if M == 1 else 3 * M - 2


def _pre(i, M):
    if M == 1:
        return i
    if i <= 0:
        return 0
    if i >= M:
        return 3 * M - 2
    return 3 * i - 1


def layer_nnz(nsd, dof, M, N, P, l0, l1):
    """Stored entries of the rows of node layers [l0,l1) of the slowest direction (closed form of
    the DMDA BOX-stencil pattern; mirrors femasm_device.h)."""
    if nsd == 3:
        return dof * dof * (_pre(l1, P) - _pre(l0, P)) * _tot(N) * _tot(M)
    return dof * dof * (_pre(l1, N) - _pre(l0, N)) * _tot(M)


def interface_counts(nsd, dof, M, N, P, ks, ke):
    """(send_count, recv_count) in doubles: the PACKED rows of one node layer + its b slice.  The rows a rank owes
    come from one element layer, so only the columns in that node layer and the one above can be non-zero: two of
    the three z-thirds (18 dof of 27 dof per interior row) are on the wire."""
    SL = P if nsd == 3 else N
    plane = M * N if nsd == 3 else M
    packed = dof * dof * 2 * (_tot(N) * _tot(M) if nsd == 3 else _tot(M))
    send = packed + plane * dof if ks > 0 else 0
    recv = packed + plane * dof if ke < SL else 0
    return send, recv


def _layer_nodes(nsd, dof, M, N, P, layer):
    """(offset in the full CSR rows of the layer, offset in the packed form, rl, prl, skip) of every node of a layer"""
    SL = P if nsd == 3 else N
    cs = _span(layer, SL)
    out, fo, po = [], 0, 0
    for j in range(N if nsd == 3 else 1):
        for i in range(M):
            crest = _span(i, M) * (_span(j, N) if nsd == 3 else 1)
            rl, prl, skip = cs * crest * dof, 2 * crest * dof, (crest * dof if layer > 0 else 0)
            out.append((fo, po, rl, prl, skip))
            fo += dof * rl
            po += dof * prl
    return out


def pack_layer_rows(nsd, dof, M, N, P, layer, full):
    """Host statement of k_iface_pack: the full CSR rows of node layer `layer` (numpy) -> the packed wire form (the
    columns in node layers `layer` and `layer`+1 only)."""
    import numpy as np
    nodes = _layer_nodes(nsd, dof, M, N, P, layer)
    out = np.empty(sum(dof * n[3] for n in nodes), dtype=full.dtype)
    for fo, po, rl, prl, skip in nodes:
        for c in range(dof):
            out[po + c * prl:po + (c + 1) * prl] = full[fo + c * rl + skip:fo + c * rl + skip + prl]
    return out


def add_packed_rows(nsd, dof, M, N, P, layer, own_full, packed):
    """Host statement of the K part of k_iface_apply (without its Dirichlet-node rule): own_full += received."""
    for fo, po, rl, prl, skip in _layer_nodes(nsd, dof, M, N, P, layer):
        for c in range(dof):
            own_full[fo + c * rl + skip:fo + c * rl + skip + prl] += packed[po + c * prl:po + (c + 1) * prl]


def exchange_interface(send, recv, rank, world):
    """rank r sends its `send` tensor to r-1 and receives `recv` from r+1.  No collective."""
    if world == 1:
        return
    ops = []
    if rank > 0 and send is not None and send.numel() > 0:
        ops.append(dist.P2POp(dist.isend, send, rank - 1))
    if rank < world - 1 and recv is not None and recv.numel() > 0:
        ops.append(dist.P2POp(dist.irecv, recv, rank + 1))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def exchange_interface_start(send, recv, rank, world):
    """Non-blocking form: enqueue the send to r-1 / receive from r+1 behind the work already queued on the current
    stream and return the requests (RCCL moves the data on its own stream while later kernels run)."""
    if world == 1:
        return []
    ops = []
    if rank > 0 and send is not None and send.numel() > 0:
        ops.append(dist.P2POp(dist.isend, send, rank - 1))
    if rank < world - 1 and recv is not None and recv.numel() > 0:
        ops.append(dist.P2POp(dist.irecv, recv, rank + 1))
    return dist.batch_isend_irecv(ops) if ops else []


def exchange_interface_finish(reqs):
    for w in reqs or []:
        w.wait()
