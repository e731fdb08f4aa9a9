"""
Post-fit statistics (the part of tapqir/utils/stats.py:89-259 that `tapqir fit` needs): credible
intervals of the variational posteriors (scipy), spot probabilities, classification scores against
simulated labels, ``<name>_params.tpqr`` / ``<name>_summary.csv`` (/ ``.mat``).

Runs once per fit on the host; rastergram plotting and the SNR / chi2 columns of the reference are not
reproduced (outside the hot path).
"""

import logging
from pathlib import Path

import numpy as np
import torch
from scipy import stats as sps

logger = logging.getLogger(__name__)


def gamma_interval(loc, beta, CI):
    """Gamma(concentration = loc*beta, rate = beta): (LL, UL, mean)."""
    conc, rate = (loc * beta).double().cpu().numpy(), beta.double().cpu().numpy()
    d = sps.gamma(conc, scale=1 / rate)
    ll, ul = d.interval(CI)
    return torch.as_tensor(ll), torch.as_tensor(ul), (loc.double().cpu())


def affine_beta_interval(mean, size, low, high, CI):
    mean, size = mean.double().cpu(), size.double().cpu()
    c1 = (size * (mean - low) / (high - low)).numpy()
    c0 = (size * (high - mean) / (high - low)).numpy()
    d = sps.beta(a=c1, b=c0, loc=low, scale=high - low)
    ll, ul = d.interval(CI)
    return torch.as_tensor(ll), torch.as_tensor(ul), mean


def dirichlet_interval(conc, CI):
    conc = conc.double().cpu()
    d = sps.beta(a=conc.numpy(), b=(conc.sum(-1, keepdim=True) - conc).numpy())
    ll, ul = d.interval(CI)
    return torch.as_tensor(ll), torch.as_tensor(ul), conc / conc.sum(-1, keepdim=True)


def save_stats(model, path, CI=0.95, save_matlab=False):
    import pandas as pd

    global_params = model._global_params
    ll_col, ul_col = f"{int(100 * CI)}% LL", f"{int(100 * CI)}% UL"
    summary = pd.DataFrame(index=global_params, columns=["Mean", ll_col, ul_col])
    logger.info("- credible intervals & spot probabilities")
    ci_stats = model.compute_params(CI)
    for param in global_params:
        for col, key in (("Mean", "Mean"), (ll_col, "LL"), (ul_col, "UL")):
            v = ci_stats[param][key]
            summary.loc[param, col] = v.item() if v.ndim == 0 else v.tolist()

    data = model.data
    if data.labels is not None:
        from sklearn.metrics import confusion_matrix, matthews_corrcoef, precision_score, recall_score

        on = data.is_ontarget.cpu()
        pred = (ci_stats["z_map"][on] > 0).numpy().ravel().astype(int)
        true = np.asarray(data.labels["z"]).ravel().astype(int)
        with np.errstate(divide="ignore", invalid="ignore"):
            summary.loc["MCC", "Mean"] = matthews_corrcoef(true, pred)
        summary.loc["Recall", "Mean"] = recall_score(true, pred, zero_division=0)
        summary.loc["Precision", "Mean"] = precision_score(true, pred, zero_division=0)
        tn, fp, fn, tp = confusion_matrix(true, pred, labels=(0, 1)).ravel()
        for k, v in (("TN", tn), ("FP", fp), ("FN", fn), ("TP", tp)):
            summary.loc[k, "Mean"] = v
    pspec = ci_stats["p_specific"][data.is_ontarget.cpu()]
    summary.loc["p(specific)", "Mean"] = float(pspec.mean()) if pspec.numel() else 0.0
    model.summary = summary

    if path is not None:
        path = Path(path)
        torch.save(ci_stats, path / f"{model.name}_params.tpqr")
        logger.info(f"Parameters were saved in {path / f'{model.name}_params.tpqr'}")
        if save_matlab:
            from scipy.io import savemat

            mat = {}
            for param, field in ci_stats.items():
                if isinstance(field, dict):
                    mat[param] = {s: np.asarray(v) for s, v in field.items()}
                else:
                    mat[param] = np.asarray(field)
            savemat(path / f"{model.name}_params.mat", mat)
        summary.to_csv(path / f"{model.name}_summary.csv")
        logger.info(f"Summary statistics were saved in {path / f'{model.name}_summary.csv'}")
