"""
Generate golden vectors from the importable fragment of the reference.

Run ONLY in the build container (needs /root/reference):

    python tests/golden/make_golden.py

It loads ``/root/reference/tapqir/distributions/util.py`` by file path (the only
module of the reference's hot path that imports without pyro/funsor/pykeops,
SURVEY.md section 8c) and stores inputs + outputs of its functions in
``tests/golden/util_golden.npz``.  The .npz holds data only; no reference source
travels with the repo.
"""

import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/tapqir/distributions/util.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "util_golden.npz")


def main():
    spec = importlib.util.spec_from_file_location("ref_util", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    torch.set_default_dtype(torch.float64)
    g = torch.Generator().manual_seed(20261003)
    out = {}

    # --- probs_m / truncated_poisson_probs / probs_theta / expand_offtarget ---
    lam = torch.tensor([0.15, 0.5, 1.0, 2.5, 1e-3])
    out["lamda"] = lam.numpy()
    for K in (1, 2, 3, 4):
        out[f"probs_m_K{K}"] = ref.probs_m(lam, K).numpy()
        out[f"trpois_K{K}"] = ref.truncated_poisson_probs(lam, K).numpy()
        out[f"probs_theta_K{K}"] = ref.probs_theta(K, torch.device("cpu")).double().numpy()
    pi = torch.tensor([[0.85, 0.15], [0.4, 0.6]])
    out["pi"] = pi.numpy()
    out["expand_offtarget"] = ref.expand_offtarget(pi).numpy()

    # --- gaussian_spots: canonical + random stacks --------------------
    P = 14
    h = torch.full((2, 3, 1, 2), 3000.0)
    w = torch.full((2, 3, 1, 2), 1.4)
    x = torch.zeros(2, 3, 1, 2)
    y = torch.zeros(2, 3, 1, 2)
    tl = torch.full((2, 3, 1, 1, 2), 6.5)
    out["gs_canon"] = ref.gaussian_spots(h, w, x, y, tl, P).numpy()

    for tag, (N, F, C, K, P) in {"a": (3, 4, 1, 2, 14), "b": (2, 3, 2, 3, 20), "c": (2, 2, 1, 1, 9)}.items():
        h = 500 + 4000 * torch.rand(N, F, C, K, generator=g)
        w = 0.75 + 1.5 * torch.rand(N, F, C, K, generator=g)
        x = (P + 1) * (torch.rand(N, F, C, K, generator=g) - 0.5)
        y = (P + 1) * (torch.rand(N, F, C, K, generator=g) - 0.5)
        tl = (P - 1) / 2 + torch.rand(N, F, C, 1, 2, generator=g) - 0.5
        m = (torch.rand(N, F, C, K, generator=g) > 0.4).double()
        for name, v in dict(h=h, w=w, x=x, y=y, tl=tl, m=m).items():
            out[f"gs_{tag}_{name}"] = v.numpy()
        out[f"gs_{tag}_P"] = np.array(P)
        out[f"gs_{tag}_out"] = ref.gaussian_spots(h, w, x, y, tl, P).numpy()
        out[f"gs_{tag}_out_m"] = ref.gaussian_spots(h, w, x, y, tl, P, m).numpy()

    # orientation probe: x=+2, y=0 must peak at columns 8-9, rows 6-7
    one = torch.ones(1)
    out["gs_orient"] = ref.gaussian_spots(
        one * 1000, one * 1.4, one * 2.0, one * 0.0, torch.full((1, 2), 6.5), 14
    ).numpy()

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
