"""Find the intermittent ~80 ms stall: per-step host enqueue time and GPU time over many steps."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from tapqir_amd.models.engine import CosmosEngine
from tapqir_amd.models.cosmos import initial_values
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate
dev = torch.device("cuda", 0)
class _M: K, device = 2, dev
data = simulate(_M, 400, 1000, 1, 14, seed=1000, params=TEST_PARAMS)
eng = CosmosEngine(data, K=2, device=dev, seed=7)
eng.layout.set_constrained(eng.params, initial_values(eng, data))
for _ in range(5): eng.step()
torch.cuda.synchronize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
host = []
ev[0].record()
for i in range(N):
    t0 = time.perf_counter()
    eng.step()
    ev[i + 1].record()
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
import statistics
print("host enqueue ms: median %.3f max %.3f (at %d)" % (statistics.median(host) * 1e3, max(host) * 1e3, host.index(max(host))))
print("gpu step ms:     median %.3f max %.3f (at %d)" % (statistics.median(gpu), max(gpu), gpu.index(max(gpu))))
big = [(i, round(g, 2), round(host[i] * 1e3, 2)) for i, g in enumerate(gpu) if g > 3 * statistics.median(gpu)]
print("outliers (step, gpu ms, host ms):", big[:20])
