"""
Golden vectors of ONE SVI evaluation (inputs and expected outputs) for the reference's canonical smoke configuration
(test/test_tapqir.py:20-50: N=2, F=5, C=1, P=14, K=2; pi=.15, lamda=.15, proximity=.2, gain=7, offset=90, height=3000,
background=150, width=1.4) and for the crosstalk model on a 2-channel variant.

    python tests/golden/make_step_golden.py        (run in the repository root)

The expected outputs come from the float64 oracle (``oracle/``, itself checked against explicit brute-force enumeration);
Pyro is not installed in this image, so no reference-produced ELBO exists ("parity unpinned").  The
fixture freezes the oracle's answers so that later changes of the oracle or of the kernels are caught; data only.
"""

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import fp32_latents, make_dataset, make_oracle, oracle_grads  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_golden.npz")


def one(tag, out, crosstalk):
    K = 2
    d = make_dataset(N=2, F=5, C=2 if crosstalk else 1, P=14, K=K, seed=0)
    o = make_oracle(d, K, perturb=0.3, seed=1, crosstalk=crosstalk)
    nd, fd = torch.arange(2), torch.arange(5)
    lat32, base = fp32_latents(o, nd, fd, seed=3)
    elbo, grads = oracle_grads(o, nd, fd, base)
    out[f"{tag}/images"] = d.images.numpy()
    out[f"{tag}/xy"] = d.xy.numpy()
    out[f"{tag}/is_ontarget"] = d.is_ontarget.numpy()
    out[f"{tag}/offset_samples"] = d.offset.samples.numpy()
    out[f"{tag}/offset_weights"] = d.offset.weights.numpy()
    for n, u in o.params.items():
        out[f"{tag}/param/{n}"] = u.detach().numpy()
    for n, v in lat32.items():
        out[f"{tag}/latent/{n}"] = v.numpy()
    for n, v in base.items():
        out[f"{tag}/base/{n}"] = v.numpy()
    out[f"{tag}/elbo"] = np.float64(elbo)
    for n, g in grads.items():
        out[f"{tag}/grad/{n}"] = g.numpy()
    for n in ("ll", "L", "T", "W", "E"):
        out[f"{tag}/term/{n}"] = o.last_terms[n].detach().numpy()
    # posterior read-out on three particles (cosmos.py:609-672)
    parts = []
    torch.manual_seed(5)
    with torch.no_grad():
        for _ in range(3):
            lat = o.sample_guide(o.params, nd, fd)
            parts.append({k: v.float().double() for k, v in lat.items()})
        z, t = o.compute_probs(nd, fd, parts)
    for i, lat in enumerate(parts):
        for n in ("pi", "lamda", "proximity", "x", "y"):
            out[f"{tag}/particle{i}/{n}"] = lat[n].numpy()
    out[f"{tag}/z_probs"] = z.numpy()
    out[f"{tag}/theta_probs"] = t.numpy()


def main():
    out = {}
    one("cosmos", out, False)
    one("crosstalk", out, True)
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays; cosmos elbo", out["cosmos/elbo"], "crosstalk elbo", out["crosstalk/elbo"])


if __name__ == "__main__":
    main()
