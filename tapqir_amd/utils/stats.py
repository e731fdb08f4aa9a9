"""
Post-fit statistics (the part of tapqir/utils/stats.py:89-259 that `tapqir fit` needs): credible
intervals of the variational posteriors (scipy), spot probabilities, classification scores against
simulated labels, ``<name>_params.tpqr`` / ``<name>_summary.csv`` (/ ``.mat``).

Runs once per fit: credible intervals (scipy, host), SNR and chi2 (stats.py:29-86, 166-193; ``tq_snr_chi2`` on the device
through ``CosmosEngine.snr_chi2``), plot ranges, classification scores and p(specific) (194-226).  The rastergram PNGs of the
reference (105-124) are not drawn.
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


def quantile(x, q):
    """Linear-interpolation quantile of a 1-D tensor (pyro.ops.stats.quantile semantics)."""
    xs = torch.sort(x.double().flatten())[0]
    pos = q * (xs.numel() - 1)
    lo, hi = int(np.floor(pos)), int(np.ceil(pos))
    return xs[lo] + (xs[hi] - xs[lo]) * (pos - lo)


def hpdi(x, prob):
    """Narrowest interval holding ``prob`` of the samples (pyro.ops.stats.hpdi semantics)."""
    xs = torch.sort(x.double().flatten())[0]
    n = xs.numel()
    k = int(prob * n)
    left, right = xs[: n - k], xs[k:]
    i = int(torch.argmin(right - left))
    return left[i], right[i]


def snr_and_chi2(data, height, width, x, y, target_locs, background, gain, offset_mean, offset_var, P, theta_probs=None):
    """Signal-to-noise ratio of each spot and chi2 of the fitted image (same arguments as tapqir/utils/stats.py:29-86;
    ``theta_probs`` is accepted and unused there too), evaluated by ``tq_snr_chi2`` on the HIP device.

    data (..., C, P, P); height / width / x / y (K, ..., Q); target_locs (..., C, 2); background (..., C) with Q = C:
    signal_k = sum_ij (D - b - offset_mean) N_k(i, j),  noise = sqrt(offset_var + b gain),  SNR = signal / noise;
    chi2 = mean_ij (D - ideal - offset_mean)^2 / ideal,  ideal = b + sum_k h_k N_k."""
    import ctypes as C

    from tapqir_amd import _lib
    from tapqir_amd.exceptions import HipExtensionError

    if data.device.type != "cuda":
        raise HipExtensionError("snr_and_chi2 runs on the HIP device only (no CPU fallback)")
    f32, dev = torch.float32, data.device
    K, lead = height.shape[0], data.shape[:-2]
    U = int(torch.Size(lead).numel())
    flat = lambda t: t.to(dev, f32).expand((K,) + lead).reshape(K, U).contiguous()
    img = data.to(f32).contiguous()
    xy = target_locs.to(dev, f32).expand(lead + (2,)).contiguous()
    h, w, xx, yy = flat(height), flat(width), flat(x), flat(y)
    b = background.to(dev, f32).expand(lead).reshape(U).contiguous()
    snr = torch.empty(K, U, dtype=f32, device=dev)
    chi2 = torch.empty(U, dtype=f32, device=dev)
    a = _lib.SnrArgs()
    p = _lib.ptr
    a.images, a.xy, a.height, a.width, a.x, a.y, a.background = p(img), p(xy), p(h), p(w), p(xx), p(yy), p(b)
    a.snr, a.chi2, a.U, a.P, a.K = p(snr), p(chi2), U, int(P), K
    a.gain, a.offset_mean, a.offset_var = float(gain), float(offset_mean), float(offset_var)
    _lib.check(_lib.load().tq_snr_chi2(C.byref(a), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "tq_snr_chi2")
    return snr.reshape((K,) + lead), chi2.reshape(lead)


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
    # plot ranges (stats.py:126-142)
    tmask = ci_stats["theta_probs"] > 0.5
    hmax = float(np.percentile(ci_stats["height"]["Mean"][tmask].numpy(), 99)) if bool(tmask.any()) else 1.0
    bmax = float(np.percentile(ci_stats["background"]["Mean"].numpy().ravel(), 99))
    for name, (lo, hi) in {"height": (-0.03 * hmax, 1.3 * hmax), "width": (0.5, 2.5), "x": (-9, 9), "y": (-9, 9),
                           "background": (-0.03 * bmax, 1.3 * bmax)}.items():
        ci_stats[name]["vmin"], ci_stats[name]["vmax"] = lo, hi
    if getattr(data, "time1", None) is not None:
        ci_stats["time1"] = data.time1
    if getattr(data, "ttb", None) is not None:
        ci_stats["ttb"] = data.ttb
    model.params = ci_stats

    logger.info("- SNR and Chi2-test")
    K, Q = model.K, model.Q
    # one kernel over all units on the device that holds the images (stats.py:166-182 loops over the AOIs on the host)
    snr, chi2 = model.engine.snr_chi2(data.offset.mean, data.offset.var)
    snr, chi2 = snr.double().cpu(), chi2.double().cpu()
    for q in range(Q):
        sel = snr[..., q][ci_stats["theta_probs"][..., q] > 0.5]
        summary.loc[f"SNR_{q}", "Mean"] = float(sel.mean()) if sel.numel() else float("nan")
    cmax = float(quantile(chi2.flatten(), 0.99))
    ci_stats["chi2"] = {"values": chi2.float(), "vmin": -0.03 * cmax, "vmax": 1.3 * cmax}

    if data.labels is not None:
        from sklearn.metrics import confusion_matrix, matthews_corrcoef, precision_score, recall_score

        on = data.is_ontarget.cpu()
        pred = (ci_stats["z_map"][on] > 0).numpy().ravel().astype(int)
        # labels of the on-target AOIs only (stats.py:196: labels["z"][: model.data.N]; read_glimpse stacks the
        # off-target labels behind them when both label files are given)
        true_z = np.asarray(data.labels["z"])[: data.N]
        true = true_z.ravel().astype(int)
        with np.errstate(divide="ignore", invalid="ignore"):
            summary.loc["MCC", "Mean"] = matthews_corrcoef(true, pred)
        summary.loc["Recall", "Mean"] = recall_score(true, pred, zero_division=0)
        summary.loc["Precision", "Mean"] = precision_score(true, pred, zero_division=0)
        tn, fp, fn, tp = confusion_matrix(true, pred, labels=(0, 1)).ravel()
        for k, v in (("TN", tn), ("FP", fp), ("FN", fn), ("TP", tp)):
            summary.loc[k, "Mean"] = v
        # z_map at the truly specific AOI-frames: median and highest-density interval (stats.py:214-226)
        zmap_on = (ci_stats["z_map"][on] > 0).long()
        samples = torch.masked_select(zmap_on, torch.from_numpy(true_z) > 0)
        if samples.numel():
            lo, hi = hpdi(samples, CI)
            summary.loc["p(specific)", "Mean"] = float(quantile(samples, 0.5))
            summary.loc["p(specific)", ll_col], summary.loc["p(specific)", ul_col] = float(lo), float(hi)
        else:
            summary.loc["p(specific)", "Mean"] = summary.loc["p(specific)", ll_col] = summary.loc["p(specific)", ul_col] = 0.0
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
