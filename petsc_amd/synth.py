"""Synthetic driver-side inputs, generated on the GPU with torch (plumbing, not the product).

These restate what the reference DRIVERS build before they call the assembly hot path, so that
bench.py and the GPU tests can create full-size inputs directly in HBM:
  * uniform_coords      -- DMDASetUniformCoordinates(da,0,1,0,1,0,1)   (P3:663, E2:578, S3:1198)
  * props_e2 / props_s3 -- the GaussPointCoefficients fill loops        (E2:618-670, S3:1243-1363)
Layouts are exactly the reference structs (E2:29-35 stride 24, S3:51-64 stride 400).
"""
import math

import torch

G = 0.57735026919
_HS = [(-1, -1, -1), (-1, 1, -1), (1, 1, -1), (1, -1, -1), (-1, -1, 1), (-1, 1, 1), (1, 1, 1), (1, -1, 1)]
_HD = [(0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1), (1, 0, 1)]
_QS = [(-1, -1), (-1, 1), (1, 1), (1, -1)]
_QD = [(0, 0), (0, 1), (1, 1), (1, 0)]


def uniform_coords(nsd, M, N, P=1, k0=0, k1=None, device="cuda", dtype=torch.float64):
    """Node coordinates of slab layers [k0,k1) of the slowest direction, interleaved, flat."""
    if nsd == 3:
        k1 = P if k1 is None else k1
        hx, hy, hz = 1.0 / (M - 1), 1.0 / (N - 1), 1.0 / (P - 1)
        x = 0.0 + hx * torch.arange(M, device=device, dtype=dtype)
        y = 0.0 + hy * torch.arange(N, device=device, dtype=dtype)
        z = 0.0 + hz * torch.arange(k0, k1, device=device, dtype=dtype)
        c = torch.empty((k1 - k0, N, M, 3), device=device, dtype=dtype)
        c[..., 0] = x[None, None, :]
        c[..., 1] = y[None, :, None]
        c[..., 2] = z[:, None, None]
        return c.reshape(-1)
    k1 = N if k1 is None else k1
    hx, hy = 1.0 / (M - 1), 1.0 / (N - 1)
    x = 0.0 + hx * torch.arange(M, device=device, dtype=dtype)
    y = 0.0 + hy * torch.arange(k0, k1, device=device, dtype=dtype)
    c = torch.empty((k1 - k0, M, 2), device=device, dtype=dtype)
    c[..., 0] = x[None, :]
    c[..., 1] = y[:, None]
    return c.reshape(-1)


def perturb_coords(coords, nsd, M, N, P=1, k0=0, amp=0.2, seed=12345):
    """'General geometry' variant: interior nodes moved by a smooth deterministic offset of
    amp*h so that no element is a parallelepiped (same formula as oracle.perturb_coords)."""
    dev = coords.device
    if nsd == 3:
        c = coords.reshape(-1, N, M, 3).clone()
        nl = c.shape[0]
        k = torch.arange(k0, k0 + nl, device=dev)[:, None, None]
        j = torch.arange(N, device=dev)[None, :, None]
        i = torch.arange(M, device=dev)[None, None, :]
        interior = (i > 0) & (i < M - 1) & (j > 0) & (j < N - 1) & (k > 0) & (k < P - 1)
        h = [1.0 / (M - 1), 1.0 / (N - 1), 1.0 / (P - 1)]
    else:
        c = coords.reshape(-1, M, 2).clone()
        nl = c.shape[0]
        j = torch.arange(k0, k0 + nl, device=dev)[:, None]
        i = torch.arange(M, device=dev)[None, :]
        interior = (i > 0) & (i < M - 1) & (j > 0) & (j < N - 1)
        h = [1.0 / (M - 1), 1.0 / (N - 1)]
    ph = (seed % 97) * 0.01
    x = c.clone()
    for d in range(nsd):
        second = x[..., (d + 2) % nsd] if nsd == 3 else x[..., d]
        arg = 7.0 * x[..., (d + 1) % nsd] + 5.0 * second + ph + d
        c[..., d] += torch.where(interior, amp * h[d] * torch.sin(arg), torch.zeros_like(arg))
    return c.reshape(-1)


def _hex_node_coords(c4, n):
    """coords of local node n of every element; c4: [L+1, N, M, 3] -> [L, N-1, M-1, 3]"""
    di, dj, dk = _HD[n]
    L, Nn, Mn = c4.shape[0] - 1, c4.shape[1], c4.shape[2]
    return c4[dk:dk + L, dj:dj + Nn - 1, di:di + Mn - 1, :]


def _hex_N(xi):
    return [0.125 * (1.0 + s[0] * xi[0]) * (1.0 + s[1] * xi[1]) * (1.0 + s[2] * xi[2]) for s in _HS]


def _face_points():
    pts = []
    s0, s1 = (-1, 1, -1, 1), (-1, -1, 1, 1)
    for f in range(6):
        axis, side = f // 2, (1.0 if f % 2 == 0 else -1.0)
        a0 = 1 if axis == 0 else 0
        a1 = 1 if axis == 2 else 2
        for q in range(4):
            x = [0.0, 0.0, 0.0]
            x[axis] = side
            x[a0] = s0[q] * G
            x[a1] = s1[q] * G
            pts.append(x)
    return pts


def props_s3(M, N, P, coords, e0=0, e1=None, ck0=0, E=90.0, nu=0.28, stride=400):
    """GaussPointCoefficients of element layers [e0,e1); coords holds node layers from ck0.
    Returns a flat device tensor [(e1-e0)*(N-1)*(M-1)*stride]."""
    e1 = P - 1 if e1 is None else e1
    dev = coords.device
    c4 = coords.reshape(-1, N, M, 3)[e0 - ck0:e1 - ck0 + 1]
    L, my, mx = e1 - e0, N - 1, M - 1
    out = torch.zeros((L, my, mx, stride), device=dev, dtype=torch.float64)
    xn = [_hex_node_coords(c4, n) for n in range(8)]
    pi = math.pi

    def interp(xi):
        Nn = _hex_N(xi)
        pos = torch.zeros_like(xn[0])
        for n in range(8):
            pos = pos + Nn[n] * xn[n]
        return pos

    for p in range(8):
        pos = interp([_HS[p][0] * G, _HS[p][1] * G, _HS[p][2] * G])
        out[..., 3 * p:3 * p + 3] = pos
        x, y, z = pos[..., 0], pos[..., 1], pos[..., 2]
        den = 2 * nu * nu + nu - 1
        out[..., 96 + p] = E
        out[..., 104 + p] = nu
        out[..., 160 + p] = (E * pi * pi * torch.cos(pi * x) * torch.cos(pi * z) * torch.sin(pi * y)) / (2 * den)
        out[..., 168 + p] = (E * pi * pi * torch.cos(pi * y) * torch.sin(pi * x) * torch.cos(pi * z)) / (2 * den)
        out[..., 176 + p] = (E * pi * pi * torch.sin(pi * x) * torch.sin(pi * y) * torch.sin(pi * z)
                             * (3 * nu - 2)) / den
    for p, xi in enumerate(_face_points()):
        pos = interp(xi)
        out[..., 24 + 3 * p:24 + 3 * p + 3] = pos
        x, y, z = pos[..., 0], pos[..., 1], pos[..., 2]
        out[..., 112 + p] = E
        out[..., 136 + p] = nu
        # gx, gy stay 0 (u_x = u_y = 0); gz = grad u_z
        out[..., 328 + 3 * p] = pi * torch.cos(pi * x) * torch.sin(pi * y) * torch.sin(pi * z)
        out[..., 328 + 3 * p + 1] = pi * torch.sin(pi * x) * torch.cos(pi * y) * torch.sin(pi * z)
        out[..., 328 + 3 * p + 2] = pi * torch.sin(pi * x) * torch.sin(pi * y) * torch.cos(pi * z)
    return out.reshape(-1)


def props_e2(M, N, coords, e0=0, e1=None, ck0=0, E=1.0, nu=0.33):
    """E2 GaussPointCoefficients (stride 24) of element rows [e0,e1)."""
    e1 = N - 1 if e1 is None else e1
    dev = coords.device
    c3 = coords.reshape(-1, M, 2)[e0 - ck0:e1 - ck0 + 1]
    L, mx = e1 - e0, M - 1
    out = torch.zeros((L, mx, 24), device=dev, dtype=torch.float64)
    xn = [c3[dj:dj + L, di:di + mx, :] for (di, dj) in _QD]
    for p in range(4):
        xi, eta = _QS[p][0] * G, _QS[p][1] * G
        pos = torch.zeros_like(xn[0])
        for n in range(4):
            pos = pos + (0.25 * (1.0 + _QS[n][0] * xi) * (1.0 + _QS[n][1] * eta)) * xn[n]
        out[..., 2 * p:2 * p + 2] = pos
        out[..., 8 + p] = E
        out[..., 12 + p] = nu
    return out.reshape(-1)


def props_enriched(M, N, P, coords, n_enr=1, e0=0, e1=None, ck0=0, E=90.0, nu=0.28, base=408):
    """Config 5 records for bench.py and the full-size tests: the R3 record (props_s3 + enrich_type = 1, n_enrich =
    n_enr at every node) followed by 64 doubles per family: psi at the 8 nodes, at the 8 Gauss points, grad psi at
    the Gauss points, psi at the 24 face points.  Families: smooth polynomials / trigonometric functions (the
    reference has no enrichment functions of its own)."""
    e1 = P - 1 if e1 is None else e1
    stride = base + 64 * n_enr
    dev = coords.device
    s3 = props_s3(M, N, P, coords, e0=e0, e1=e1, ck0=ck0, E=E, nu=nu, stride=400).reshape(-1, 400)
    nel = s3.shape[0]
    out = torch.zeros((nel, stride), device=dev, dtype=torch.float64)
    out[:, :400] = s3
    itype = torch.int32 if base == 408 else torch.int64
    flags = torch.ones((nel, 16), device=dev, dtype=itype)
    flags[:, 8:] = n_enr
    out[:, 400:base] = flags.view(torch.float64).reshape(nel, base - 400)
    c4 = coords.reshape(-1, N, M, 3)[e0 - ck0:e1 - ck0 + 1]
    xn = [_hex_node_coords(c4, n).reshape(-1, 3) for n in range(8)]
    gp = s3[:, 0:24].reshape(nel, 8, 3)
    gpb = s3[:, 24:96].reshape(nel, 24, 3)

    def psi(m, x, y, z):
        if m == 0:
            return x * x + 0.5 * y * z + 0.25 * z, 2 * x, 0.5 * z, 0.5 * y + 0.25
        if m == 1:
            return torch.sin(2 * x) * y, 2 * torch.cos(2 * x) * y, torch.sin(2 * x), torch.zeros_like(x)
        return z * z * x + y, z * z, torch.ones_like(x), 2 * z * x

    for m in range(n_enr):
        o = base + 64 * m
        for a in range(8):
            out[:, o + a] = psi(m, xn[a][:, 0], xn[a][:, 1], xn[a][:, 2])[0]
        v = psi(m, gp[:, :, 0], gp[:, :, 1], gp[:, :, 2])
        out[:, o + 8:o + 16] = v[0]
        out[:, o + 16:o + 40] = torch.stack([v[1], v[2], v[3]], dim=-1).reshape(nel, 24)
        out[:, o + 40:o + 64] = psi(m, gpb[:, :, 0], gpb[:, :, 1], gpb[:, :, 2])[0]
    return out.reshape(-1)
