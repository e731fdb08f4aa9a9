"""
Scientific end-to-end check on the GPU: fit simulated data with the reference's canonical parameters
(test/test_tapqir.py:20-50) and compare the classification of target-specific spots with the simulated labels and the
fitted global parameters with the generating ones.
"""

import math

import pytest
import torch

from tapqir_amd.models import models
from tapqir_amd.utils.dataset import save
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

pytestmark = pytest.mark.gpu


def mcc(pred, truth):
    tp = float((pred & truth).sum()); tn = float((~pred & ~truth).sum())
    fp = float((pred & ~truth).sum()); fn = float((~pred & truth).sum())
    den = math.sqrt((tp + fp) * (tp + fn) * (tn + fp) * (tn + fn))
    return (tp * tn - fp * fn) / den if den else 0.0


@pytest.mark.parametrize("name", ["cosmos", "crosstalk"])
def test_fit_recovers_labels_and_globals(tmp_path, name):
    xt = name == "crosstalk"
    N, F = (10, 100) if xt else (20, 200)
    params = dict(TEST_PARAMS, alpha=[[0.9, 0.1], [0.2, 0.8]]) if xt else TEST_PARAMS
    d = simulate(2, N, F, 2 if xt else 1, 14, seed=3, params=params)
    save(d, tmp_path)
    m = models[name](S=1, K=2, device="cuda", dtype="float")
    m.load(tmp_path)
    m.init(lr=0.005, nbatch_size=N, fbatch_size=F)
    m.run(3000, progress_bar=lambda x: x)
    assert m.iter == 3000  # no NaN-recovery restart happened (model.py:220-232 would rewind the counter)
    cp = {n: v.detach().cpu() for n, v in m.engine.layout.constrained(m.engine.params).items()}
    assert abs(float(cp["gain_loc"]) - 7.0) < 0.7
    assert float(cp["proximity_loc"]) < 0.6 and (cp["pi_mean"][:, 1] < 0.3).all() and (cp["lamda_loc"] < 0.45).all()
    if xt:
        assert (cp["alpha_mean"] - torch.tensor([[0.9, 0.1], [0.2, 0.8]])).abs().max() < 0.05
    z = m.z_probs
    zmap = (z[..., 1] > 0.5) if z.dim() == 4 else (z > 0.5)
    truth = torch.as_tensor(d.labels["z"]).bool()
    assert mcc(zmap[: N // 2].cpu(), truth) > 0.93
