"""
The subsample of a minibatch step drawn on the device (include/tapqir_hip.h: next_ndx / next_fdx; CosmosEngine.step_subsampled):
the single-launch minibatch step of step t also draws `randperm(Nt)[:nb]`, `randperm(F)[:fb]` of step t + 1 (pyro.plate's
subsample, tapqir/models/cosmos.py:194-208) as the nb / fb smallest of Nt / F Philox keys.

  * a step on a device-drawn subsample IS the step on the same indices handed over by the host;
  * the law: distinct indices, every AOI / frame equally likely, fresh draws every step, reproducible from (seed, step).
"""

import pytest
import torch

from helpers import CosmosEngine, make_dataset, make_oracle, oracle_to_engine

pytestmark = pytest.mark.gpu


def _used(eng):
    """Indices the step just enqueued ran on (the launch writes the OTHER slot)."""
    st = eng._sub
    return st["slots"][1 - st["turn"]].clone()


@pytest.mark.parametrize("K", [1, 2])
def test_device_subsample_steps_equal_host_fed_steps(K):
    N, F, nb, fb = 12, 40, 5, 16
    d = make_dataset(N=N, F=F, K=K)
    o = make_oracle(d, K, perturb=0.2)
    a = CosmosEngine(d, K=K, device="cuda:0", seed=9)
    b = CosmosEngine(d, K=K, device="cuda:0", seed=9)
    for e in (a, b):
        oracle_to_engine(o, e)
    g = torch.Generator().manual_seed(1)
    for it in range(8):
        assert a.step_subsampled(nb, fb, g)
        used = _used(a).cpu()
        nd, fd = used[:nb].long(), used[N:N + fb].long()
        assert nd.unique().numel() == nb and fd.unique().numel() == fb and int(nd.max()) < N and int(fd.max()) < F
        b.step(nd, fd)
        for e in (a, b):
            e.join()
        torch.cuda.synchronize()
        ea, eb = float(a.elbo_out[0]), float(b.elbo_out[0])
        assert abs(ea - eb) <= 1e-9 * abs(eb), (it, ea, eb)
    va, vb = a.named("params"), b.named("params")
    for n in va:
        assert torch.equal(va[n], vb[n]), n


def test_device_subsample_law():
    N, F, nb, fb, steps = 12, 40, 5, 16, 400
    d = make_dataset(N=N, F=F, K=1)
    o = make_oracle(d, 1)
    eng = CosmosEngine(d, K=1, device="cuda:0", seed=21)
    oracle_to_engine(o, eng)
    twin = CosmosEngine(d, K=1, device="cuda:0", seed=21)
    oracle_to_engine(o, twin)
    g = torch.Generator().manual_seed(0)
    cn, cf = torch.zeros(N), torch.zeros(F)
    seen = []
    for it in range(steps):
        assert eng.step_subsampled(nb, fb, g)
        used = _used(eng).cpu()
        nd, fd = used[:nb].long(), used[N:N + fb].long()
        assert nd.unique().numel() == nb and fd.unique().numel() == fb
        cn[nd] += 1
        cf[fd] += 1
        seen.append((tuple(sorted(nd.tolist())), tuple(sorted(fd.tolist()))))
    eng.join()
    torch.cuda.synchronize()
    assert torch.isfinite(eng.params).all()
    # every AOI / frame equally likely (binomial: 5 sigma)
    for c, p in ((cn, nb / N), (cf, fb / F)):
        sd = (steps * p * (1 - p)) ** 0.5
        assert float((c - steps * p).abs().max()) < 5 * sd, (c, steps * p, sd)
    assert len(set(seen)) > 0.95 * steps  # fresh draws every step
    # reproducible from (seed, step): a second engine with the same seed and step count draws the same subsamples
    g2 = torch.Generator().manual_seed(0)
    for it in range(5):
        assert twin.step_subsampled(nb, fb, g2)
        used = _used(twin).cpu()
        assert (tuple(sorted(used[:nb].tolist())), tuple(sorted(used[N:N + fb].tolist()))) == seen[it]
