"""Host time to enqueue minibatch steps vs their device time (reference default minibatch 10 x 512)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tapqir_amd.models.cosmos import initial_values
from tapqir_amd.models.engine import CosmosEngine
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

dev = torch.device("cuda", 0)
class _M: K, device = 2, dev
data = simulate(_M, 400, 1000, 1, 14, seed=1000, params=TEST_PARAMS)
eng = CosmosEngine(data, K=2, device=dev, seed=7)
eng.layout.set_constrained(eng.params, initial_values(eng, data))
g = torch.Generator().manual_seed(0)
idx = [(torch.randperm(400, generator=g)[:10], torch.randperm(1000, generator=g)[:512]) for _ in range(200)]
for nd, fd in idx[:20]:
    eng.step(nd, fd)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for nd, fd in idx:
        eng.step(nd, fd)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e6 * (t1 - t0) / len(idx):.1f} us/step, drained after {1e6 * (t2 - t0) / len(idx):.1f} us/step", flush=True)
# device-resident indices: no staging copies
didx = [(a.to(dev, torch.int32), b.to(dev, torch.int32)) for a, b in idx]
for rep in range(2):
    t0 = time.perf_counter()
    for nd, fd in didx:
        eng.step(nd, fd)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"device indices: enqueue {1e6 * (t1 - t0) / len(idx):.1f} us/step, drained after {1e6 * (t2 - t0) / len(idx):.1f} us/step", flush=True)
