"""Feasibility probe: does the (HBM-bound) fused pixel + per-unit launch overlap with the (VALU-bound) local sampling
launch when both are in flight on two streams?  Two engines with separate buffers; times alone, back to back, and
concurrently."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tapqir_amd.models.cosmos import initial_values
from tapqir_amd.models.engine import CosmosEngine
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate
dev = torch.device("cuda", 0)
class _M: K, device = 2, dev
N = int(os.environ.get("AOIS", 400))
data = simulate(_M, N, 1000, 1, 14, seed=1000, params=TEST_PARAMS)
engs = []
for i in range(2):
    e = CosmosEngine(data, K=2, device=dev, seed=7 + i)
    e.layout.set_constrained(e.params, initial_values(e, data))
    e.pixel_mode, e.fuse_unit = 0, True
    for _ in range(3):
        e.step()
    e.join()
    engs.append(e)
e1, e2 = engs
a1 = e1.make_args(); a1.fuse_adam, a1.pixel_mode, a1.last_step = 1, 2, None
e1.call("cosmos_sample_globals", a1); e1.call("cosmos_sample_locals", a1)
a2 = e2.make_args()
e2.call("cosmos_sample_globals", a2)
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
h1, h2 = C.c_void_p(s1.cuda_stream), C.c_void_p(s2.cuda_stream)
fused = lambda: e1.lib.tq_cosmos_pixel_unit(C.byref(a1), h1)
sample = lambda st: e2.lib.tq_cosmos_sample_locals(C.byref(a2), st)
def timed(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
saved = e1.params.clone(), e1.exp_avg.clone(), e1.exp_avg_sq.clone()
for rep in range(2):
    tf = timed(fused); ts = timed(lambda: sample(h2)); tb = timed(lambda: (fused(), sample(h1)))
    tc = timed(lambda: (fused(), sample(h2)))
    print(f"fused alone {tf:.1f} us, sampling alone {ts:.1f} us, same stream {tb:.1f} us, two streams {tc:.1f} us")
    e1.params.copy_(saved[0]); e1.exp_avg.copy_(saved[1]); e1.exp_avg_sq.copy_(saved[2])
