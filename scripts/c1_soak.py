import sys, torch
sys.path.insert(0, ".")
from tapqir_amd.models.cosmos import initial_values
from tapqir_amd.models.engine import CosmosEngine
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate
dev = torch.device("cuda", 0)
class _M: K, device = 1, dev
data = simulate(_M, 50, 100, 1, 14, seed=1000, params=TEST_PARAMS)
e = CosmosEngine(data, K=1, device=dev, seed=7)
e.layout.set_constrained(e.params, initial_values(e, data))
g = torch.Generator().manual_seed(0)
for it in range(30000):
    if it % 7 == 3:
        e.step(torch.randperm(50, generator=g)[:5], torch.randperm(100, generator=g)[:64])
    else:
        e.step()
e.join(); torch.cuda.synchronize()
print("c1 soak 30000 steps (small full batches in one launch, every 7th a 5 x 64 minibatch): ELBO", float(e.elbo_out[0]), "finite", bool(torch.isfinite(e.params).all()), "gave up waiting", int(e._sync[63]))
