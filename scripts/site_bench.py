"""Time the local guide-site kernel with and without drawing (diagnostic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tapqir_amd.models.cosmos import initial_values
from tapqir_amd.models.engine import CosmosEngine
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

dev = torch.device("cuda", 0)


class _M:
    K, device = 2, dev


data = simulate(_M, 400, 1000, 1, 14, seed=1000, params=TEST_PARAMS)
eng = CosmosEngine(data, K=2, device=dev, seed=7)
eng.layout.set_constrained(eng.params, initial_values(eng, data))
import time

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
state = sys.argv[2] if len(sys.argv) > 2 else None  # file with trained parameters: loaded if present, else written
if state and os.path.exists(state):
    eng.params.copy_(torch.load(state).to(dev))
    nsteps = 0
for _ in range(nsteps):
    eng.step()
torch.cuda.synchronize()
if state and not os.path.exists(state):
    torch.save(eng.params.cpu(), state)
t0 = time.perf_counter()
for _ in range(50):
    eng.step()
torch.cuda.synchronize()
print(f"after {nsteps} steps: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/step")
cp = eng.layout.constrained(eng.params)
print("size quantiles", torch.quantile(cp["size"].flatten()[:1000000].float(), torch.tensor([0.01, 0.5, 0.99], device=dev)).tolist(),
      "w_size", torch.quantile(cp["w_size"].flatten()[:1000000].float(), torch.tensor([0.01, 0.5, 0.99], device=dev)).tolist(),
      "h alpha", torch.quantile((cp["h_loc"] * cp["h_beta"]).flatten()[:1000000].float(), torch.tensor([0.01, 0.5, 0.99], device=dev)).tolist())
for draw in (1, 0):
    a = eng.make_args(draw_locals=bool(draw))
    for _ in range(3):
        eng.call("cosmos_sample_locals", a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.call("cosmos_sample_locals", a)
    e1.record()
    e1.synchronize()
    print(f"sample_locals draw={draw}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
