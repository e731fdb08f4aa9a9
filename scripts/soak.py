"""Soak run: 10 000 full-batch steps, 40 000 minibatch steps with fresh subsamples, 300 steps alternating between the two kinds
(pending tails handed from one kind to the other), 20 000 steps on device-drawn subsamples, 20 000 minibatch steps with a 50-value
offset histogram (20 units per workgroup, tail claimed by the last block); everything must stay finite, no launch may stall and no
workgroup may ever have given up waiting for a flag (sync word 63)."""
import os, sys, time, torch
sys.path.insert(0, ".")
from tapqir_amd.models.cosmos import initial_values
from tapqir_amd.models.engine import CosmosEngine
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate
dev = torch.device("cuda", 0)
class _M: K, device = 2, dev
data = simulate(_M, 400, 1000, 1, 14, seed=1000, params=TEST_PARAMS)
e = CosmosEngine(data, K=2, device=dev, seed=7)
e.layout.set_constrained(e.params, initial_values(e, data))
t0 = time.time()
for it in range(10000):
    e.step()
e.join(); torch.cuda.synchronize()
print("full batch 10000 steps", round(time.time() - t0, 2), "s; ELBO", float(e.elbo_out[0]), "finite", bool(torch.isfinite(e.params).all()), "fused", e.fuse_unit, "pixel_mode", e.pixel_mode)
g = torch.Generator().manual_seed(0)
t0 = time.time()
for it in range(40000):
    e.step(torch.randperm(400, generator=g)[:10], torch.randperm(1000, generator=g)[:512])
    if it % 5000 == 4999:
        e.join(); print("  minibatch", it + 1, float(e.elbo_out[0]), bool(torch.isfinite(e.params).all()), flush=True)
# alternate kinds of steps (pending tails of either kind handed over)
for it in range(300):
    if it % 3 == 0: e.step()
    else: e.step(torch.randperm(400, generator=g)[:10], torch.randperm(1000, generator=g)[:512])
e.join(); torch.cuda.synchronize()
print("minibatch 40000 + mixed 300 steps", round(time.time() - t0, 2), "s; ELBO", float(e.elbo_out[0]), "finite", bool(torch.isfinite(e.params).all()))
# device-drawn subsamples (the launch of step t draws the subsample of step t + 1)
t0 = time.time()
for it in range(20000):
    assert e.step_subsampled(10, 512, g)
e.join(); torch.cuda.synchronize()
print("device-subsampled 20000 steps", round(time.time() - t0, 2), "s; ELBO", float(e.elbo_out[0]), "finite", bool(torch.isfinite(e.params).all()))
print("workgroups that gave up waiting for a flag:", int(e._sync[63]))
# offset histogram: 257 workgroups of 20 units, the tail on the block dispatched last, two flags
from tapqir_amd.utils.dataset import CosmosDataset
s_ = torch.arange(70.0, 120.0)
w_ = torch.minimum(s_ - 69.0, 120.0 - s_)
h = CosmosEngine(CosmosDataset(data.images, data.xy, data.is_ontarget, offset_samples=s_, offset_weights=w_ / w_.sum()), K=2, device=dev, seed=7)
h.layout.set_constrained(h.params, initial_values(h, data))
t0 = time.time()
bad = 0
for it in range(20000):
    assert h.step_subsampled(10, 512, g)
    if it % 100 == 99:
        h.join()
        bad += int(not torch.isfinite(h.elbo_out).all())
h.join(); torch.cuda.synchronize()
print("histogram minibatch 20000 steps", round(time.time() - t0, 2), "s; ELBO", float(h.elbo_out[0]), "finite", bool(torch.isfinite(h.params).all()),
      "non-finite losses seen", bad, "workgroups that gave up waiting:", int(h._sync[63]))
