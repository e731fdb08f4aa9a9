"""End-to-end fit on simulated data: global parameter recovery and spot classification against the labels."""
import argparse
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def mcc(pred, truth):
    tp = float((pred & truth).sum()); tn = float((~pred & ~truth).sum())
    fp = float((pred & ~truth).sum()); fn = float((~pred & truth).sum())
    den = ((tp + fp) * (tp + fn) * (tn + fp) * (tn + fn)) ** 0.5
    return (tp * tn - fp * fn) / den if den else 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="cosmos")
    ap.add_argument("--N", type=int, default=20)
    ap.add_argument("--F", type=int, default=200)
    ap.add_argument("--iters", type=int, default=4000)
    ap.add_argument("--nbatch", type=int, default=0)
    ap.add_argument("--fbatch", type=int, default=0)
    ap.add_argument("--lr", type=float, default=0.005)
    a = ap.parse_args()
    from tapqir_amd.models import models
    from tapqir_amd.utils.dataset import save
    from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

    xt = a.model == "crosstalk"
    params = dict(TEST_PARAMS, alpha=[[0.9, 0.1], [0.2, 0.8]]) if xt else TEST_PARAMS
    d = simulate(2, a.N, a.F, 2 if xt else 1, 14, seed=3, params=params)
    tmp = tempfile.mkdtemp()
    save(d, tmp)
    m = models[a.model](S=1, K=2, device="cuda", dtype="float")
    m.load(tmp)
    m.init(lr=a.lr, nbatch_size=a.nbatch or a.N, fbatch_size=a.fbatch or a.F)
    t0 = time.time()
    done = 0
    while done < a.iters:
        m.run(500, progress_bar=lambda x: x)
        done += 500
        cp = {n: v.detach().cpu() for n, v in m.engine.layout.constrained(m.engine.params).items()}
        msg = (f"iter {done:5d}  -ELBO {m.iter_loss:.6g}  gain {float(cp['gain_loc']):.3f}  pi {cp['pi_mean'][:, 1].tolist()}  "
               f"lamda {cp['lamda_loc'].tolist()}  proximity {float(cp['proximity_loc']):.3f}")
        if xt:
            msg += f"  alpha {cp['alpha_mean'].tolist()}"
        print(msg, flush=True)
    print(f"{a.iters} iterations in {time.time() - t0:.1f} s")
    z = m.z_probs
    zmap = (z[..., 1] > 0.5) if z.dim() == 4 else (z > 0.5)
    truth = torch.as_tensor(d.labels["z"]).bool()
    pred = zmap[: a.N // 2].cpu()
    print(f"MCC {mcc(pred, truth):.4f}  truth-positive fraction {truth.float().mean():.3f}  predicted {pred.float().mean():.3f}")


if __name__ == "__main__":
    main()
