"""How long the tail workgroup of a pipelined full-batch step lives inside the sampling launch (TQ_MB_STAMPS build)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tapqir_amd.models.cosmos import initial_values
from tapqir_amd.models.engine import CosmosEngine
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate
dev = torch.device("cuda", 0)
class _M: K, device = 2, dev
data = simulate(_M, 400, 1000, 1, 14, seed=1000, params=TEST_PARAMS)
e = CosmosEngine(data, K=2, device=dev, seed=7)
e.layout.set_constrained(e.params, initial_values(e, data))
e.pixel_mode, e.fuse_unit = 0, True
for _ in range(50): e.step()
torch.cuda.synchronize()
st = e._sync[4:32].cpu().view(torch.int64)
t = lambda i: (int(st[i]) - int(st[8])) / 100.0
print("tail workgroup of the previous step inside the sampling launch (us after its start): row / group sums loaded %.1f, gsum written %.1f, "
      "global sites done %.1f, Adam of the per-AOI / global parameters %.1f, global draws of this step %.1f; (about) the last sampling workgroup %.1f"
      % (t(9) if int(st[9]) > int(st[8]) else float("nan"), t(12) if int(st[12]) > int(st[8]) else float("nan"), t(7), t(13), t(11), t(10)))
