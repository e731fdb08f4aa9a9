"""Minibatch steps with a 50-value offset histogram until the first non-finite loss (or STEPS steps): which step, which parameters."""
import os, sys, time, torch
sys.path.insert(0, ".")
from tapqir_amd.models.cosmos import initial_values
from tapqir_amd.models.engine import CosmosEngine
from tapqir_amd.utils.dataset import CosmosDataset
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate
dev = torch.device("cuda", 0)
KK = int(os.environ.get("K", 2))
class _M: K, device = KK, dev
XT = os.environ.get("MODEL") == "crosstalk"
NA, NF = int(os.environ.get("AOIS", 400)), int(os.environ.get("FRAMES", 1000))
data = simulate(_M, NA, NF, 2 if XT else 1, int(os.environ.get("P", 14)), seed=1000,
                params=dict(TEST_PARAMS, alpha=[[0.9, 0.1], [0.2, 0.8]]) if XT else TEST_PARAMS)
s_ = torch.arange(70.0, 120.0)
w_ = torch.minimum(s_ - 69.0, 120.0 - s_)
if os.environ.get("OFFSETS") == "wide":  # offsets reaching above the dimmest pixels: masked offsets in most pixels' loops
    s_ = torch.arange(70.0, 330.0, 4.0)
    w_ = torch.exp(-0.5 * ((s_ - 90.0) / 60.0) ** 2)
    data = CosmosDataset(data.images, data.xy, data.is_ontarget, offset_samples=s_, offset_weights=(w_ / w_.sum()).float())
elif os.environ.get("OFFSETS", "hist") == "hist":
    data = CosmosDataset(data.images, data.xy, data.is_ontarget, offset_samples=s_, offset_weights=w_ / w_.sum())
h = CosmosEngine(data, K=KK, device=dev, seed=int(os.environ.get("SEED", 7)), crosstalk=XT)
if XT:
    from tapqir_amd.models.crosstalk import crosstalk_initial_values as xt_initial_values
    h.layout.set_constrained(h.params, xt_initial_values(h, data))
else:
    h.layout.set_constrained(h.params, initial_values(h, data))
FULL = os.environ.get("FULL") == "1"
g = torch.Generator().manual_seed(0)
steps, every = int(os.environ.get("STEPS", 20000)), int(os.environ.get("EVERY", 50))
host_draw = os.environ.get("HOST_DRAW") == "1"
prev = h.params.clone(); prev_m = h.exp_avg.clone(); prev_v = h.exp_avg_sq.clone()
for it in range(steps):
    if FULL:
        h.step()
    elif host_draw or XT:
        h.step(torch.randperm(NA, generator=g)[:10], torch.randperm(NF, generator=g)[:512])
    else:
        assert h.step_subsampled(10, 512, g)
    if it + 1 >= int(os.environ.get("DENSE_FROM", 10**9)):
        every = 1
    if it % every == every - 1:
        h.join()
        if every == 1 and bool(torch.isfinite(h.params).all()):
            prev, prev_m, prev_v = h.params.clone(), h.exp_avg.clone(), h.exp_avg_sq.clone()
        if not bool(torch.isfinite(h.elbo_out).all()) or not bool(torch.isfinite(h.params).all()):
            bad = {n: int((~torch.isfinite(v)).sum()) for n, v in h.named("params").items() if not bool(torch.isfinite(v).all())}
            print(f"non-finite at step {it + 1}: loss {float(h.elbo_out[0])}, parameters {bad}, gave up waiting {int(h._sync[63])}")
            for n, v in h.named("params").items():
                if n in bad:
                    idx = (~torch.isfinite(v)).nonzero()[:4].tolist()
                    print("  ", n, "first bad indices", idx)
            st_ = h._sub["slots"][1 - h._sub["turn"]]
            for n_, v_ in h.named("params").items():
                if n_ in bad and v_.dim() == 4:
                    _, an, fr, _ = (~torch.isfinite(v_)).nonzero()[0].tolist()
                    print("   AOI", an, "in this batch:", an in st_[:10].tolist(), "frame", fr, "in this batch:", fr in st_[400:400 + 512].tolist(),
                          "last_step of the unit", int(h._last_step[an * 1000 + fr]), "adam_step", h.adam_step)
                    ai, bi = st_[:10].tolist().index(an) if an in st_[:10].tolist() else -1, st_[400:912].tolist().index(fr) if fr in st_[400:912].tolist() else -1
                    if ai >= 0 and bi >= 0:
                        i = ai * 512 + bi
                        Bq = 5120
                        print("   position", i, "lat", [float(x) for x in h.lat.view(-1, Bq)[:, i]])
                        print("   pix", [float(x) for x in h.pix.view(-1, Bq)[:, i]])
                        print("   site", [[float(x) for x in h.site.view(5, 9, Bq)[r, :, i]] for r in range(5)])
                        u = an * 1000 + fr
                        for nm, buf in (("params before", prev), ("params after", h.params), ("exp_avg after", h.exp_avg), ("exp_avg_sq after", h.exp_avg_sq)):
                            print("   ", nm, [float(buf[r * 400000 + u]) for r in range(18)])
                        print("    prev exp_avg", [float(prev_m[r * 400000 + u]) for r in range(18)])
                        print("    prev exp_avg_sq", [float(prev_v[r * 400000 + u]) for r in range(18)])
                    break
            # the step's workspace: which rows of the latents / site terms / pixel results are not finite, and for which units
            B = 10 * 512
            K, M = 2, 4
            for name, buf, rows in (("lat", h.lat, 1 + 4 * K), ("pix", h.pix, M + 2 + 4 * K)):
                v = buf[: rows * B].view(rows, B)
                badrows = [(r, (~torch.isfinite(v[r])).nonzero().flatten()[:3].tolist()) for r in range(rows) if not bool(torch.isfinite(v[r]).all())]
                print("  ", name, "non-finite rows (row, first positions):", badrows)
                if name == "pix" and badrows:
                    i = badrows[0][1][0]
                    print("   unit position", i, "lat", [float(h.lat[: (1 + 4 * K) * B].view(-1, B)[r, i]) for r in range(1 + 4 * K)])
                    print("   pix", [float(v[r, i]) for r in range(rows)])
                    st = h._sub["slots"][1 - h._sub["turn"]]
                    n, f = int(st[i // 512]), int(st[400 + i % 512])
                    t = data.images[n, f, 0]
                    print("   AOI", n, "frame", f, "tile min/max", float(t.min()), float(t.max()), "gain", float(h.globals[0]))
            sv = h.site[: 5 * (1 + 4 * K) * B].view(5, 1 + 4 * K, B)
            print("   site rows non-finite:", [(a_, b_) for a_ in range(5) for b_ in range(1 + 4 * K) if not bool(torch.isfinite(sv[a_, b_]).all())])
            break
else:
    h.join()
    print(f"{steps} steps finite; loss {float(h.elbo_out[0])}; gave up waiting {int(h._sync[63])}")
