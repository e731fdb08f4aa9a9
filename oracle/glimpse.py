"""
Oracle restatement of ``tapqir/imscroll/glimpse_reader.py`` (TEST INFRASTRUCTURE ONLY): parsing of the imscroll
files and the per-frame / per-AOI extraction loop of ``read_glimpse``, in plain numpy (no pandas, no GPU).

Parity unpinned: the reference's tests hold no fixture for this path (test/test_imscroll.py covers the kinetics
helpers only) and the module cannot be imported here (it imports pyro through tapqir.utils.dataset), so this file
follows the reference's text, cited per function, and is checked against closed-form cases in tests/test_glimpse.py.

One deliberate reading: ``GlimpseDataset.__getitem__`` (glimpse_reader.py:168-186) returns ``int16_array + 2**15``.
Under the value-based casting of the numpy releases the reference was written for this is an int32 array with values
0..65535; numpy >= 2 raises OverflowError for the same expression.  The oracle (and the HIP kernel) implement the
former, which is the documented intent (unsigned 16-bit camera counts).
"""

from collections import OrderedDict

import numpy as np
import torch
from scipy.io import loadmat


# ---- imscroll metadata (glimpse_reader.py:57-164) ------------------------------------------------------------------
def read_header(folder):
    """glimpse_reader.py:62-66: header.mat -> {field: squeezed array}."""
    vid = loadmat(str(folder) + "/header.mat")["vid"]
    return {name: np.squeeze(vid[0, 0][i]) for i, name in enumerate(vid.dtype.names)}


def read_aoiinfo(path):
    """glimpse_reader.py:79-101 -> (aoi numbers (N,), x (N,), y (N,), frame of the first record) with x, y converted
    to 0-based pixel coordinates.  Columns of the table: frame, ave, y, x, pixnum, aoi."""
    try:
        mat = loadmat(str(path))
    except ValueError:
        mat = np.loadtxt(str(path))
    if isinstance(mat, dict):
        table = mat["aoiinfo2"] if "aoiinfo2" in mat else mat["aoifits"]["aoiinfo2"][0, 0]
    else:
        table = mat
    table = np.asarray(table, dtype=np.float64).reshape(-1, 6)
    return table[:, 5].astype(np.int64), table[:, 3] - 1, table[:, 2] - 1, table[:, 0]


def cumulative_drift(frames, dxy, aoiinfo_frame):
    """glimpse_reader.py:103-112.  ``frames`` (F,) sorted frame numbers, ``dxy`` (F, 2) per-frame drift (dx, dy).
    Rows after the aoiinfo frame hold the running sum of the drift since that frame, rows before it minus the drift
    still to come up to (and including) that frame; the row OF the aoiinfo frame keeps its raw value."""
    frames = np.asarray(frames)
    out = np.array(dxy, dtype=np.float64, copy=True)
    pa = int(np.searchsorted(frames, aoiinfo_frame))
    if pa >= len(frames) or frames[pa] != aoiinfo_frame:
        raise ValueError(f"aoiinfo frame {aoiinfo_frame} is not in the driftlist")
    acc = np.zeros(2)
    for p in range(pa + 1, len(frames)):
        acc = acc + dxy[p]
        out[p] = acc
    acc = np.zeros(2)
    for p in range(pa - 1, -1, -1):
        acc = acc + (-dxy[p + 1])
        out[p] = acc
    return out


def read_driftlist(path):
    """glimpse_reader.py:68-74: columns frame, dy, dx -> (frames int (F,), dxy (F, 2) as (dx, dy))."""
    dl = loadmat(str(path))["driftlist"][:, :3]
    return dl[:, 0].astype(np.int64), np.stack([dl[:, 2], dl[:, 1]], -1).astype(np.float64)


def spotpicker_labels(path, aois, frames):
    """glimpse_reader.py:118-150: binary labels from imscroll ``Intervals.CumulativeIntervalArray`` rows
    (code, first frame, last frame, ..., aoi): codes -2, 0, 2 mark absence, -3, 1, 3 presence."""
    rows = loadmat(str(path))["Intervals"]["CumulativeIntervalArray"][0, 0]
    sp = np.zeros((len(aois), len(frames)))
    for row in rows:
        aoi, first, last = int(row[-1]), int(row[1]), int(row[2])
        sel = (np.asarray(aois)[:, None] == aoi) & (np.asarray(frames)[None, :] >= first) & (np.asarray(frames)[None, :] <= last)
        if row[0] in (-2.0, 0.0, 2.0):
            sp[sel] = 0
        elif row[0] in (-3.0, 1.0, 3.0):
            sp[sel] = 1
    lab = np.zeros(sp.shape, dtype=[("aoi", int), ("frame", int), ("z", bool), ("spotpicker", float)])
    lab["aoi"] = np.asarray(aois).reshape(-1, 1)
    lab["frame"] = np.asarray(frames)
    lab["spotpicker"] = sp
    lab["z"] = sp
    return lab


# ---- frames -----------------------------------------------------------------------------------------------------------
def decode_frame(raw, H, W):
    """glimpse_reader.py:181-186: H*W big-endian int16 + 2**15 (see the module docstring)."""
    return np.frombuffer(raw, dtype=">i2", count=H * W).astype(np.int64).reshape(H, W) + 2 ** 15


def read_frame(folder, header, frame):
    """glimpse_reader.py:176-186; ``frame`` is the 1-based frame number of the driftlist."""
    H, W = int(header["height"]), int(header["width"])
    number = np.atleast_1d(header["filenumber"])[frame - 1]
    offset = int(np.atleast_1d(header["offset"])[frame - 1])
    with open(f"{folder}/{number}.glimpse", "rb") as fid:
        fid.seek(offset)
        return decode_frame(fid.read(2 * H * W), H, W)


# ---- the extraction loop (glimpse_reader.py:358-392) ------------------------------------------------------------------
def extract_frame(img, raw_xy_f, P, images_f, target_xy_f):
    """One frame: ``raw_xy_f`` (N, 2) -> ``images_f`` (N, P, P) accumulated, ``target_xy_f`` (N, 2) set."""
    for n in range(raw_xy_f.shape[0]):
        shiftx = round(raw_xy_f[n, 0] - 0.5 * (P - 1))
        shifty = round(raw_xy_f[n, 1] - 0.5 * (P - 1))
        images_f[n] += img[shifty:shifty + P, shiftx:shiftx + P]  # ValueError when the window leaves the frame
        target_xy_f[n, 0] = raw_xy_f[n, 0] - shiftx
        target_xy_f[n, 1] = raw_xy_f[n, 1] - shifty


def count_offsets(img, offset_x, offset_y, offset_P, counts):
    """glimpse_reader.py:362-369: value counts of the offset region, accumulated in ``counts`` {value: count}."""
    region = img[offset_y:offset_y + offset_P, offset_x:offset_x + offset_P]
    values, cnt = np.unique(region, return_counts=True)
    for v, k in zip(values, cnt):
        counts[int(v)] = counts.get(int(v), 0) + int(k)
    return np.median(region) if region.size else float("nan")


def bin_hist(samples, weights, s):
    """glimpse_reader.py:22-38: keep the first sample, merge every following run of ``s`` samples into its middle
    one, and a shorter last run into its own middle one.  Weights accumulate in torch's default dtype."""
    q, r = divmod(len(samples) - 1, s)
    n = 1 + q + (1 if r else 0)
    out_s = torch.zeros(n, dtype=torch.int)
    out_w = torch.zeros(n)
    out_s[0], out_w[0] = samples[0], weights[0]
    for b in range(q):
        lo = 1 + b * s
        out_s[1 + b] = samples[lo + s // 2]
    for i in range(s):
        for b in range(q):
            out_w[1 + b] += weights[1 + b * s + i]
    if r:
        out_s[-1] = samples[1 + q * s + r // 2]
        out_w[-1] = weights[1 + q * s:].sum()
    return out_s, out_w


def finish_offsets(counts, min_data, bin_size):
    """glimpse_reader.py:413-436: sorted value counts -> (samples int32, weights) after the low sentinel, the
    normalisation, folding of the top 0.5 % into the last kept sample and ``bin_hist``."""
    counts = OrderedDict(sorted(counts.items()))
    samples = np.array(list(counts.keys()))
    weights = np.array(list(counts.values()))
    if min_data <= samples[0]:
        samples = np.insert(samples, 0, min_data - 1)
        weights = np.insert(weights, 0, 1)
    weights = weights / weights.sum()
    high = weights.cumsum() > 0.995
    high_w = weights[high].sum()
    samples, weights = samples[~high], weights[~high]
    weights[-1] += high_w
    return bin_hist(torch.tensor(samples, dtype=torch.int), torch.tensor(weights), bin_size)


def read_glimpse(**kwargs):
    """The whole of read_glimpse (glimpse_reader.py:304-470) minus plotting and saving, as a dict of the arrays the
    reference hands to CosmosDataset: images int64 (Nt, F, C, P, P), xy float64 (Nt, F, C, 2), is_ontarget,
    offset_samples, offset_weights, labels, time1, ttb."""
    kwargs = dict(kwargs)
    P, C = kwargs.pop("P"), kwargs.pop("num-channels")
    channels = kwargs.pop("channels")
    offset_P, bin_size = kwargs.pop("offset-P"), kwargs.pop("bin-size")
    dtypes = ["ontarget"] + (["offtarget"] if kwargs["use-offtarget"] else [])
    counts, medians = {}, []
    data = {d: [] for d in dtypes}
    txy = {d: [] for d in dtypes}
    labels = {d: [] for d in dtypes}
    time1, ttb = [], []
    for c in range(C):
        ch = channels[c]
        header = read_header(ch["glimpse-folder"])
        frames, dxy = read_driftlist(ch["driftlist"])
        aoi = {d: read_aoiinfo(ch[f"{d}-aoiinfo"]) for d in dtypes}
        a_on = aoi["ontarget"]
        first = int(a_on[3][list(a_on[0]).index(1)])  # .at[1, "frame"]: the record of AOI number 1
        cum = cumulative_drift(frames, dxy, first)
        ttb_all = np.atleast_1d(header["ttb"])
        if kwargs["frame-range"]:
            keep = (frames >= int(kwargs["frame-start"])) & (frames <= int(kwargs["frame-end"]))
            frames, cum, ttb_all = frames[keep], cum[keep], ttb_all[keep]
        time1.append(float(header["time1"]))
        ttb.append(ttb_all)
        F = len(frames)
        raw = {d: np.stack([aoi[d][1], aoi[d][2]], -1)[:, None, :] + cum[None] for d in dtypes}
        for d in dtypes:
            data[d].append(np.zeros((len(aoi[d][0]), F, P, P), dtype="int"))
            txy[d].append(np.zeros((len(aoi[d][0]), F, 2)))
            lab = None
            if kwargs.get("labels") and ch.get(f"{d}-labels") is not None:
                lab = spotpicker_labels(ch[f"{d}-labels"], aoi[d][0], frames)
            labels[d].append(lab)
        for f, frame in enumerate(frames):
            img = read_frame(ch["glimpse-folder"], header, int(frame))
            medians.append(count_offsets(img, kwargs["offset-x"], kwargs["offset-y"], offset_P, counts))
            for d in dtypes:
                extract_frame(img, raw[d][:, f], P, data[d][c][:, f], txy[d][c][:, f])
        for d in dtypes:
            assert (txy[d][c] > 0.5 * P - 1).all() and (txy[d][c] < 0.5 * P).all()
    min_data = min(np.stack(data[d], -3).min() for d in dtypes)
    samples, weights = finish_offsets(counts, min_data, bin_size)
    lab_out = []
    for d in dtypes:
        if all(l is not None for l in labels[d]):
            lab_out.append(np.stack(labels[d], -1))
    return dict(
        images=torch.tensor(np.concatenate([np.stack(data[d], -3) for d in dtypes], 0)),
        xy=torch.tensor(np.concatenate([np.stack(txy[d], -2) for d in dtypes], 0)),
        is_ontarget=torch.cat([torch.full((data[d][0].shape[0],), d == "ontarget", dtype=torch.bool) for d in dtypes]),
        offset_samples=samples, offset_weights=weights, offset_medians=np.array(medians),
        labels=np.concatenate(lab_out, 0) if lab_out else None,
        time1=torch.as_tensor(time1), ttb=torch.as_tensor(np.array(ttb)).T)
