"""Run the cosmos fit step by step and report the first non-finite quantity (diagnostic)."""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tapqir_amd.models import models
from tapqir_amd.utils.dataset import save
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

import argparse

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="cosmos")
ap.add_argument("--N", type=int, default=20)
ap.add_argument("--F", type=int, default=200)
ap.add_argument("--iters", type=int, default=4000)
ap.add_argument("--nbatch", type=int, default=0)
ap.add_argument("--fbatch", type=int, default=0)
ap.add_argument("--seed", type=int, default=3)
args = ap.parse_args()
N, F = args.N, args.F
xt = args.model == "crosstalk"
C_ = 2 if xt else 1
d = simulate(2, N, F, C_, 14, seed=args.seed, params=dict(TEST_PARAMS, alpha=[[0.9, 0.1], [0.2, 0.8]]) if xt else TEST_PARAMS)
tmp = tempfile.mkdtemp()
save(d, tmp)
m = models[args.model](S=1, K=2, device="cuda", dtype="float")
m.load(tmp)
nbs, fbs = args.nbatch or N, args.fbatch or F
m.init(lr=0.005, nbatch_size=nbs, fbatch_size=fbs)
eng = m.engine
K, B = 2, nbs * fbs * C_
names_lat = ["b", "h0", "h1", "w0", "w1", "x0", "x1", "y0", "y1"]
for it in range(args.iters):
    prev = eng.params.clone()
    ndx, fdx = m._subsample()
    eng.step(ndx, fdx)
    eng.join()
    torch.cuda.synchronize()
    elbo = float(eng.elbo_out[0])
    ok = torch.isfinite(eng.params).all().item() and elbo == elbo and abs(elbo) != float("inf")
    if not ok:
        print("first non-finite at iteration", it, "elbo", elbo)
        lat = eng.lat.view(1 + 4 * K, B)
        site = eng.site.view(5, 1 + 4 * K, B)  # TQ_NSITE_STORED rows: lq, d alpha (c1), grad, d c0, grad0
        pix = eng.pix.view(-1, B)
        for r in range(lat.shape[0]):
            bad = ~torch.isfinite(lat[r])
            if bad.any():
                print("  lat", names_lat[r], int(bad.sum()))
        for j in range(6):
            for r in range(site.shape[1]):
                bad = ~torch.isfinite(site[j, r])
                if bad.any():
                    i = int(bad.nonzero()[0])
                    print(f"  site term {j} of {names_lat[r]}: {int(bad.sum())} bad; unit {i} lat={float(lat[r, i])!r} terms={[float(site[t, r, i]) for t in range(5)]}")
        for r in range(pix.shape[0]):
            bad = ~torch.isfinite(pix[r])
            if bad.any():
                print("  pix row", r, int(bad.sum()), "first unit", int(bad.nonzero()[0]))
        g = eng.layout.views(eng.params)
        for n, v in g.items():
            if not torch.isfinite(v).all():
                print("  param", n, int((~torch.isfinite(v)).sum()))
        print("  gsum", eng._gsum_buf[:8].tolist(), "globals", eng.globals[:12].tolist())
        break
else:
    print(f"no non-finite value in {args.iters} iterations ({args.model}, batch {nbs}x{fbs}); -ELBO {-float(eng.elbo_out[0]):.6g}")
