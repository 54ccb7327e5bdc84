"""The one entry point that allocates nothing (include/femasm.h: femasm_error_integrals_static) against the
entry point it is a variant of, through the C ABI: bitwise-equal sums (same kernels, same order), on two calls in a
row, the second with a smaller input -- the static scratch keeps the first call's partial sums, which must not leak
into the second result.  Collected last (test_zz_*): the round's end runs pytest -x."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_error_integrals_static_equals_error_integrals_twice_smaller_second():
    import torch
    from petsc_amd import femasm as F
    from petsc_amd import synth
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    # first call: 40 x 33 x 29 elements = 38 280 -> 150 workgroups; second: 9 x 7 x 5 = 315 -> 2 workgroups
    for kind, dof in ((F.ELASTIC3D_Q1, 3), (F.POISSON3D_Q1, 1)):
        for (M, N, P) in ((41, 34, 30), (10, 8, 6)):
            coords = synth.perturb_coords(synth.uniform_coords(3, M, N, P, device=dev), 3, M, N, P)
            plan = F.Plan(kind, M, N, P, coords=coords)
            u = torch.from_numpy(rng.uniform(-1.0, 1.0, size=M * N * P * dof)).to(dev)
            want = plan.error_integrals(u)                       # the existing entry point (plan-owned partial sums)
            sums = torch.full((2,), float("nan"), dtype=torch.float64, device=dev)
            plan.error_integrals_static(u, sums)                 # asynchronous: nothing copied back by the call
            plan.sync()
            got = sums.cpu().numpy()
            assert got[0] == want[0] and got[1] == want[1], (kind, M, got, want)
            assert want[0] > 0.0 and want[1] > 0.0
            plan.close()


def test_error_integrals_static_two_plans_on_two_streams_do_not_share_scratch():
    """Two plans of one GPU (two slabs, say) driven on two streams at once: each plan owns a slice of the GPU's static
    array, so their partial sums cannot meet (round-2 advice: one shared array raced)."""
    import torch
    from petsc_amd import femasm as F
    from petsc_amd import synth
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    plans, us, wants, sums, streams = [], [], [], [], []
    for (M, N, P) in ((70, 61, 52), (33, 90, 41)):
        coords = synth.perturb_coords(synth.uniform_coords(3, M, N, P, device=dev), 3, M, N, P)
        plan = F.Plan(F.ELASTIC3D_Q1, M, N, P, coords=coords)
        u = torch.from_numpy(rng.uniform(-1.0, 1.0, size=M * N * P * 3)).to(dev)
        wants.append(plan.error_integrals(u))
        st = torch.cuda.Stream()
        plan.set_stream(st.cuda_stream)
        plans.append(plan); us.append(u); streams.append(st)
        sums.append(torch.full((2,), float("nan"), dtype=torch.float64, device=dev))
    torch.cuda.synchronize()
    for rep in range(20):  # interleaved: the two kernels pairs overlap on the device
        for q in range(2):
            plans[q].error_integrals_static(us[q], sums[q])
    torch.cuda.synchronize()
    for q in range(2):
        got = sums[q].cpu().numpy()
        assert got[0] == wants[q][0] and got[1] == wants[q][1], (q, got, wants[q])
        plans[q].close()


def test_error_integrals_static_rejects_what_it_cannot_do():
    import torch
    from petsc_amd import femasm as F
    from petsc_amd import synth
    dev = torch.device("cuda:0")
    M = 5
    coords = synth.uniform_coords(3, M, M, M, device=dev)
    plan = F.Plan(F.POISSON3D_Q1, M, M, M, coords=coords)
    u = torch.zeros(M ** 3, dtype=torch.float64, device=dev)
    with pytest.raises(F.FemasmError):  # host result pointer: staging would allocate
        plan.error_integrals_static(u, np.zeros(2))
    plan.close()
    # a slab beyond the static scratch (more than 256^3 elements) is refused with a clear message
    M, N, P = 258, 258, 258
    big = synth.uniform_coords(3, M, N, P, device=dev)
    plan = F.Plan(F.POISSON3D_Q1, M, N, P, coords=big)
    u = torch.zeros(M * N * P, dtype=torch.float64, device=dev)
    sums = torch.zeros(2, dtype=torch.float64, device=dev)
    with pytest.raises(F.FemasmError, match="exceed the static scratch"):
        plan.error_integrals_static(u, sums)
    plan.close()
