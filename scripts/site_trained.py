"""Per-site timing of the local guide-sampling kernel in the TRAINED parameter regime (after STEPS full-batch steps of
fitting, default 4000; the parameters are cached in gpurun_out/ so that diagnostic builds time the same state)."""
import ctypes as C, os, sys
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
steps = int(os.environ.get("STEPS", 4000))
cache = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", f"trained_params_{steps}.pt")
if os.path.exists(cache):
    eng.params.copy_(torch.load(cache, weights_only=True).to(dev))
else:
    for _ in range(steps):
        eng.step()
    eng.join()
    os.makedirs(os.path.dirname(cache), exist_ok=True)
    torch.save(eng.params.cpu(), cache)
v = eng.layout.constrained(eng.params)
q = lambda t: [round(float(x), 2) for x in torch.quantile(t.flatten().float()[:4_000_000], torch.tensor([0.1, 0.5, 0.9], device=t.device))]
print("size", q(v["size"]), "w_size", q(v["w_size"]), "h alpha", q(v["h_loc"] * v["h_beta"]), "b alpha", q(v["b_loc"] * v["b_beta"]), "m_probs", q(v["m_probs"]))
names = ["b"] + [f"{s}{k}" for s in "hwxy" for k in range(2)]
a = eng.make_args()
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def timed(site, n):
    for _ in range(3):
        eng.lib.tq_cosmos_sample_locals_range(C.byref(a), site, n, None, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.lib.tq_cosmos_sample_locals_range(C.byref(a), site, n, None, st)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
out = [timed(s, 1) for s in range(9)]
print(f"trained ({steps} steps)", " ".join(f"{n}={t:.1f}" for n, t in zip(names, out)), f"sum={sum(out):.1f} all-in-one={timed(0, 9):.1f} us")
