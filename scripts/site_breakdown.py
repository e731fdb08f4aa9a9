"""Per-site timing of the local guide-sampling kernel (tq_cosmos_sample_locals_range, one site per launch) at the
initial parameters and after some full-batch steps."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tapqir_amd import _lib
from tapqir_amd.models.cosmos import initial_values
from tapqir_amd.models.engine import CosmosEngine
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

dev = torch.device("cuda", 0)


class _M:
    K, device = 2, dev


data = simulate(_M, 400, 1000, 1, 14, seed=1000, params=TEST_PARAMS)
eng = CosmosEngine(data, K=2, device=dev, seed=7)
eng.layout.set_constrained(eng.params, initial_values(eng, data))
names = ["b"] + [f"{s}{k}" for s in "hwxy" for k in range(2)]  # site order of `lat`
NSTEPS = int(os.environ.get("STEPS", 300))
for phase, nsteps in (("init", 0), (f"after {NSTEPS} steps", NSTEPS)):
    for _ in range(nsteps):
        eng.step()
    eng.join()
    a = eng.make_args()
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    out = []
    for site in range(1 + 4 * eng.K):
        for _ in range(3):
            eng.lib.tq_cosmos_sample_locals_range(C.byref(a), site, 1, None, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            eng.lib.tq_cosmos_sample_locals_range(C.byref(a), site, 1, None, st)
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1) / 20 * 1e3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.lib.tq_cosmos_sample_locals_range(C.byref(a), 0, 1 + 4 * eng.K, None, st)
    e1.record()
    e1.synchronize()
    print(phase, " ".join(f"{n}={t:.1f}" for n, t in zip(names, out)), f"sum={sum(out):.1f} all-in-one={e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
