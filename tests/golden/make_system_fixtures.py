#!/usr/bin/env python3
"""Writes tests/golden/systems.npz: inputs and assembled systems (K values, b, after the Dirichlet step) of small
meshes, produced by the CPU oracle (oracle/femasm_oracle.c, itself pinned to survey_8c.json).  SURVEY 8(c) "to
add": 2^3, 3^3, 4^3 and 5x4x3 meshes, distorted elements, per-Gauss-point E/nu, the natural-BC vector Ge.
These are regression fixtures for both the oracle (bitwise) and the HIP path (1e-13): the reference itself cannot
be built here (no PETSc), so they are not reference outputs.

    python tests/golden/make_system_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle as O  # noqa: E402


def cases():
    rng = np.random.default_rng(20261111)
    out = {}
    for name, dims, perturb in [("p3_2x2x2", (2, 2, 2), False), ("p3_3x3x3", (3, 3, 3), False),
                                ("p3_4x4x4_distorted", (4, 4, 4), True), ("p3_5x4x3_distorted", (5, 4, 3), True)]:
        M, N, P = dims
        c = O.uniform_coords(3, M, N, P)
        if perturb:
            c = O.perturb_coords(c, 3, M, N, P)
        rp, ci = O.pattern(3, 1, M, N, P)
        K, b = O.assemble_poisson3d(M, N, P, c, rp, ci)
        out[name] = dict(kind=0, dims=dims, coords=c, props=np.zeros(0), stride=0, K=K, b=b)
    for name, dims, perturb, vary in [("e2_3x3", (3, 3), False, False), ("e2_6x5_distorted_varying", (6, 5), True, True)]:
        M, N = dims
        c = O.uniform_coords(2, M, N)
        if perturb:
            c = O.perturb_coords(c, 2, M, N)
        rp, ci = O.pattern(2, 2, M, N)
        pr = O.props_e2(M - 1, N - 1, c).reshape(-1, 24)
        pr[:, 16:24] = rng.uniform(-1, 1, size=(pr.shape[0], 8))
        if vary:
            pr[:, 8:12] = rng.uniform(0.5, 2.0, size=(pr.shape[0], 4))
            pr[:, 12:16] = rng.uniform(0.1, 0.4, size=(pr.shape[0], 4))
        pr = pr.reshape(-1)
        K, b = O.assemble_elastic2d(M, N, c, pr, rp, ci)
        O.bc_e2(M, N, rp, ci, K, b)
        out[name] = dict(kind=1, dims=(M, N, 1), coords=c, props=pr, stride=24, K=K, b=b)
    for name, dims, perturb, vary, stride in [("s3_2x2x2", (2, 2, 2), False, False, 400),
                                              ("s3_3x3x3", (3, 3, 3), False, False, 400),
                                              ("s3_4x4x4_distorted", (4, 4, 4), True, False, 400),
                                              ("s3_5x4x3_distorted_varying", (5, 4, 3), True, True, 400),
                                              ("r3_4x3x3_stride416", (4, 3, 3), False, False, 416)]:
        M, N, P = dims
        c = O.uniform_coords(3, M, N, P)
        if perturb:
            c = O.perturb_coords(c, 3, M, N, P)
        rp, ci = O.pattern(3, 3, M, N, P)
        pr = O.props_s3(M - 1, N - 1, P - 1, c, stride=stride).reshape(-1, stride)
        if vary:
            pr[:, 96:104] = rng.uniform(50, 150, size=(pr.shape[0], 8))
            pr[:, 104:112] = rng.uniform(0.1, 0.4, size=(pr.shape[0], 8))
            pr[:, 112:136] = rng.uniform(50, 150, size=(pr.shape[0], 24))
            pr[:, 136:160] = rng.uniform(0.1, 0.4, size=(pr.shape[0], 24))
            pr[:, 184:400] = rng.uniform(-1, 1, size=(pr.shape[0], 216))
        pr = pr.reshape(-1)
        K, b = O.assemble_elastic3d(M, N, P, c, pr, rp, ci, stride=stride)
        O.bc_s3(M, N, P, c, rp, ci, K, b)
        out[name] = dict(kind=2, dims=dims, coords=c, props=pr, stride=stride, K=K, b=b)
    return out


if __name__ == "__main__":
    flat = {}
    for name, d in cases().items():
        flat[name + "/meta"] = np.array([d["kind"], *d["dims"], d["stride"]], dtype=np.int64)
        for k in ("coords", "props", "K", "b"):
            flat[name + "/" + k] = d[k]
    path = os.path.join(HERE, "systems.npz")
    np.savez_compressed(path, **flat)
    print(path, os.path.getsize(path), "bytes,", len(flat) // 5, "cases")
