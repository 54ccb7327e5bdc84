"""CPU tests of the host-side slab logic (petsc_amd/slab.py) and of the synthetic-input generators
(petsc_amd/synth.py) against the oracle's restatement of the reference drivers' set-up loops."""
import numpy as np
import torch


def test_dmda_slab_split_matches_the_survey_examples():
    from petsc_amd import slab
    r = slab.slab_ranges(257, 8)
    assert [b - a for a, b in r] == [33] + [32] * 7 and r[0][0] == 0 and r[-1][1] == 257
    r = slab.slab_ranges(129, 8)
    assert [b - a for a, b in r] == [17] + [16] * 7
    assert slab.element_range(0, 33) == (0, 32) and slab.element_range(33, 65) == (32, 64)
    # every element layer is assembled exactly once
    for P, p in ((129, 8), (10, 3), (5, 5)):
        seen = []
        for ks, ke in slab.slab_ranges(P, p):
            e0, e1 = slab.element_range(ks, ke)
            seen += list(range(e0, e1))
        assert seen == list(range(P - 1))


def test_layer_nnz_closed_form_matches_the_pattern(O):
    from petsc_amd import slab
    for nsd, dof, (M, N, P) in ((3, 3, (4, 3, 6)), (2, 2, (5, 7, 1)), (3, 1, (2, 2, 2))):
        rp, _ = O.pattern(nsd, dof, M, N, P)
        SL = P if nsd == 3 else N
        plane = dof * (M * N if nsd == 3 else M)
        for l0 in range(SL):
            for l1 in range(l0 + 1, SL + 1):
                assert slab.layer_nnz(nsd, dof, M, N, P, l0, l1) == rp[plane * l1] - rp[plane * l0]
    # the wire form holds two of the three z-thirds of the layer's rows (21.6 MB at 128^3 S3) + b
    s, r = slab.interface_counts(3, 3, 129, 129, 129, 17, 33)
    assert s == r == 9 * (3 * 129 - 2) ** 2 * 2 + 129 * 129 * 3
    assert abs(8 * s / 1e6 - 21.7) < 0.2
    assert slab.interface_counts(3, 3, 129, 129, 129, 0, 129) == (0, 0)


def test_synth_inputs_equal_the_oracle_driver_restatement(O):
    from petsc_amd import synth
    M, N, P = 6, 5, 7
    c = synth.uniform_coords(3, M, N, P, device="cpu")
    co = O.uniform_coords(3, M, N, P)
    assert np.array_equal(c.numpy(), co)
    cp = synth.perturb_coords(c, 3, M, N, P)
    assert np.abs(cp.numpy() - O.perturb_coords(co, 3, M, N, P)).max() < 1e-15
    pr = synth.props_s3(M, N, P, cp).numpy()
    po = O.props_s3(M - 1, N - 1, P - 1, cp.numpy())
    assert np.abs(pr - po).max() <= 1e-12 * np.abs(po).max()
    # a slab of element layers with a coordinate array that starts above layer 0
    sl = synth.props_s3(M, N, P, cp.reshape(P, -1)[1:5].reshape(-1), e0=2, e1=4, ck0=1).numpy()
    assert np.array_equal(sl, pr.reshape(P - 1, -1)[2:4].reshape(-1))
    c2 = synth.uniform_coords(2, M, N, device="cpu")
    assert np.array_equal(c2.numpy(), O.uniform_coords(2, M, N))
    p2 = synth.props_e2(M, N, synth.perturb_coords(c2, 2, M, N)).numpy()
    assert np.abs(p2 - O.props_e2(M - 1, N - 1, O.perturb_coords(O.uniform_coords(2, M, N), 2, M, N))).max() < 1e-15
    assert np.array_equal(synth.uniform_coords(3, M, N, P, 2, 5, device="cpu").numpy(),
                          co.reshape(P, -1)[2:5].reshape(-1))
