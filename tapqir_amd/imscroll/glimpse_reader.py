"""
``tapqir glimpse`` preprocessing: raw Glimpse frames + imscroll files -> ``data.tpqr``; drop-in for
tapqir/imscroll/glimpse_reader.py (``bin_hist`` 22-38, ``GlimpseDataset`` 41-216, ``read_glimpse`` 304-470).

The metadata (header.mat, driftlist, aoiinfo, spot-picker intervals) is parsed on the host into the same
``GlimpseDataset`` attributes the reference exposes.  The data-parallel part -- decoding the big-endian frames,
cutting the drift-corrected P x P window of every AOI out of every frame, counting the offset-region values --
runs on the GPU through ``tq_glimpse_extract`` (include/tapqir_hip.h): the frame bytes go to HBM exactly as they
sit in the ``.glimpse`` files, in chunks, with the next chunk read from disk while the previous one is processed.
There is no CPU fallback.  The diagnostic PNG plots of the reference (glimpse_reader.py:218-301, 354-360, 472-501)
are not produced.
"""

import logging
from collections import defaultdict
from pathlib import Path
from typing import Tuple

import numpy as np
import torch

from tapqir_amd import _lib
from tapqir_amd.utils.dataset import CosmosDataset, save

logger = logging.getLogger(__name__)

CHUNK_BYTES = 256 << 20  # frame bytes per tq_glimpse_extract call


def bin_hist(samples: torch.Tensor, weights: torch.Tensor, s: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Thin the offset histogram (glimpse_reader.py:22-38): the first sample stays; every following run of ``s``
    samples collapses onto its middle sample with the run's total weight; a shorter trailing run likewise.  The
    weights are accumulated in the default dtype, run member by run member, as the reference does."""
    q, r = divmod(len(samples) - 1, s)
    body = slice(1, 1 + q * s)
    mids = samples[body].reshape(q, s)[:, s // 2]
    acc = torch.zeros(q)
    runs = weights[body].reshape(q, s)
    for i in range(s):
        acc += runs[:, i]
    new_samples = [samples[:1].to(torch.int), mids.to(torch.int)]
    new_weights = [weights[:1].to(acc.dtype), acc]
    if r:
        tail = 1 + q * s
        new_samples.append(samples[tail + r // 2].reshape(1).to(torch.int))
        new_weights.append(weights[tail:].sum().reshape(1).to(acc.dtype))
    return torch.cat(new_samples), torch.cat(new_weights)


def _aoi_table(path):
    """The (N, 6) table frame, ave, y, x, pixnum, aoi of an aoiinfo file (glimpse_reader.py:79-96): a MATLAB file
    holding ``aoiinfo2`` or ``aoifits.aoiinfo2``, or a plain text table."""
    from scipy.io import loadmat

    try:
        mat = loadmat(path)
    except ValueError:
        return np.loadtxt(path)
    if "aoiinfo2" in mat:
        return mat["aoiinfo2"]
    return mat["aoifits"]["aoiinfo2"][0, 0]


class GlimpseDataset:
    """
    Metadata of one colour channel of a Glimpse / imscroll experiment (interface of glimpse_reader.py:41-164).

    Keyword arguments are the per-channel entries of ``.tapqir/config.yaml`` merged with the command's options:
    ``name`` (label of the channel), ``glimpse-folder`` (directory holding ``header.mat`` and the ``N.glimpse`` frame
    files), ``driftlist`` (stage drift per frame), ``ontarget-aoiinfo`` / ``offtarget-aoiinfo`` (AOI positions: MATLAB
    file or text table; the second only with ``use-offtarget``), ``frame-range`` with ``frame-start`` / ``frame-end``
    (restrict the analysis to these frames, inclusive), ``labels`` with ``ontarget-labels`` / ``offtarget-labels``
    (spot-picker interval files), ``offset-x`` / ``offset-y`` (corner of the dark region used for the camera offset).

    Attributes, named as in the reference: ``header`` (dict of the ``vid`` struct), ``aoiinfo[dtype]`` (DataFrame indexed
    by AOI number, x / y converted to 0-based pixels), ``cumdrift`` (DataFrame indexed by frame number: dx, dy summed
    relative to the frame the AOIs were picked in, ttb), ``labels[dtype]``, ``dtypes``, ``height``, ``width``, ``N``,
    ``Nc``, ``F``.
    """

    def __init__(self, c=0, **kwargs):
        import pandas as pd
        from scipy.io import loadmat

        dtypes = ["ontarget"] + (["offtarget"] if kwargs["use-offtarget"] else [])
        vid = loadmat(Path(kwargs["glimpse-folder"]) / "header.mat")["vid"]
        header = {field: np.squeeze(vid[0, 0][i]) for i, field in enumerate(vid.dtype.names)}

        dl = loadmat(kwargs["driftlist"])["driftlist"][:, :3]
        drift = pd.DataFrame({"dy": dl[:, 1], "dx": dl[:, 2]}, index=pd.Index(dl[:, 0].astype(int), name="frame"))
        drift["ttb"] = header["ttb"]

        aoiinfo = {}
        for dtype in dtypes:
            df = pd.DataFrame(np.asarray(_aoi_table(kwargs[f"{dtype}-aoiinfo"]), dtype=float).reshape(-1, 6),
                              columns=["frame", "ave", "y", "x", "pixnum", "aoi"])
            df = df.astype({"aoi": int}).set_index("aoi")
            df[["x", "y"]] -= 1  # MATLAB -> 0-based pixel coordinates
            aoiinfo[dtype] = df

        # drift accumulated relative to the frame the AOIs were picked in (glimpse_reader.py:103-112): running sums
        # forwards from the next frame, minus the drift still to come backwards; that frame itself keeps its entry
        ref = int(aoiinfo["ontarget"].at[1, "frame"])
        pos = drift.index.get_loc(ref)
        d = drift[["dx", "dy"]].to_numpy(copy=True)
        cum = d.copy()
        cum[pos + 1:] = np.cumsum(d[pos + 1:], axis=0)
        cum[:pos] = np.cumsum(-d[pos:0:-1], axis=0)[::-1]
        drift[["dx", "dy"]] = cum

        if kwargs["frame-range"]:
            drift = drift.loc[int(kwargs["frame-start"]):int(kwargs["frame-end"])]

        labels = defaultdict(lambda: None)
        for dtype in dtypes:
            if kwargs["labels"] and kwargs[f"{dtype}-labels"] is not None:
                labels[dtype] = self._spotpicker(kwargs[f"{dtype}-labels"], aoiinfo[dtype].index.values, drift.index.values)

        self.height, self.width = int(header["height"]), int(header["width"])
        self.config = kwargs
        self.header = header
        self.dtypes = dtypes
        self.aoiinfo = aoiinfo
        self.cumdrift = drift
        self.labels = labels
        self.name = kwargs["name"]
        self.c = c
        self.offset_x = kwargs["offset-x"]
        self.offset_y = kwargs["offset-y"]

    @staticmethod
    def _spotpicker(path, aois, frames):
        """Binary labels from the imscroll interval table (glimpse_reader.py:118-150): rows (code, first frame,
        last frame, ..., aoi); codes -2/0/2 = no spot, -3/1/3 = spot."""
        from scipy.io import loadmat

        lab = np.zeros((len(aois), len(frames)),
                       dtype=[("aoi", int), ("frame", int), ("z", bool), ("spotpicker", float)])
        lab["aoi"] = aois.reshape(-1, 1)
        lab["frame"] = frames
        for row in loadmat(path)["Intervals"]["CumulativeIntervalArray"][0, 0]:
            if row[0] in (-2.0, 0.0, 2.0):
                value = 0
            elif row[0] in (-3.0, 1.0, 3.0):
                value = 1
            else:
                continue
            hit = (lab["aoi"] == int(row[-1])) & (lab["frame"] >= int(row[1])) & (lab["frame"] <= int(row[2]))
            lab["spotpicker"][hit] = value
        lab["z"] = lab["spotpicker"]
        return lab

    # -- frames ----------------------------------------------------------------------------------------------------------
    def frame_location(self, frame):
        """(file path, byte offset) of the 1-based ``frame`` (glimpse_reader.py:177-180)."""
        number = np.atleast_1d(self.header["filenumber"])[frame - 1]
        return Path(self.config["glimpse-folder"]) / f"{number}.glimpse", int(np.atleast_1d(self.header["offset"])[frame - 1])

    def read_raw(self, frames, out):
        """Bytes of ``frames`` (1-based numbers) into the uint8 array ``out`` (len(frames) * H * W * 2), as stored."""
        nbytes = 2 * self.height * self.width
        i = 0
        while i < len(frames):
            path, offset = self.frame_location(int(frames[i]))
            j = i + 1  # extend the read over frames stored back to back in the same file
            while j < len(frames) and self.frame_location(int(frames[j])) == (path, offset + (j - i) * nbytes):
                j += 1
            with open(path, "rb") as fid:
                fid.seek(offset)
                got = fid.readinto(memoryview(out[i * nbytes:j * nbytes]))
            if got != (j - i) * nbytes:
                raise ValueError(f"{path}: frame {int(frames[i])} is truncated ({got} of {(j - i) * nbytes} bytes)")
            i = j

    def __getitem__(self, key):
        """The whole frame image(s) as host arrays (glimpse_reader.py:168-186), for inspection and plotting."""
        if isinstance(key, slice):
            return np.stack([self[f] for f in range(key.start, key.stop, 1 if key.step is None else key.step)], 0)
        raw = np.empty(2 * self.height * self.width, dtype=np.uint8)
        self.read_raw([key], raw)
        return raw.view(">i2").astype(np.int32).reshape(self.height, self.width) + 2 ** 15

    def __len__(self) -> int:
        return self.F

    @property
    def N(self) -> int:
        return len(self.aoiinfo["ontarget"])

    @property
    def Nc(self) -> int:
        return len(self.aoiinfo["offtarget"]) if "offtarget" in self.dtypes else 0

    @property
    def F(self) -> int:
        return len(self.cumdrift)

    def __repr__(self):
        return f"{self.__class__.__name__}(N={self.N}, Nc={self.Nc}, F={self.F})"

    __str__ = __repr__


def _finish_offsets(hist, min_data, bin_size):
    """Offset value counts -> (samples, weights) of the dataset (glimpse_reader.py:413-436)."""
    samples = np.flatnonzero(hist)
    if len(samples) == 0:
        raise ValueError("the offset region is empty: check offset-x, offset-y and offset-P against the frame size")
    counts = hist[samples]
    if min_data <= samples[0]:  # a sentinel below every data value keeps all pixels above some offset sample
        samples = np.concatenate([[min_data - 1], samples])
        counts = np.concatenate([[1], counts])
    weights = counts / counts.sum()
    keep = ~(weights.cumsum() > 0.995)  # the top 0.5 % is folded into the last kept sample
    folded = weights[~keep].sum()
    samples, weights = samples[keep], weights[keep]
    weights[-1] += folded
    return bin_hist(torch.tensor(samples, dtype=torch.int), torch.tensor(weights), bin_size)


def extract_aois(glimpse, raw_xy, P, images, target_xy, hist, status, c, offset_P, progress_bar=None,
                 chunk_bytes=None):
    """Run ``tq_glimpse_extract`` over all frames of one channel.  ``raw_xy`` (N, F, 2) float64 host array; ``images``
    int32 (N, F, C, P, P), ``target_xy`` float64 (N, F, C, 2), ``hist`` int64 [65536], ``status`` int32 [2]: device."""
    device = images.device
    if device.type != "cuda":
        raise _lib.HipExtensionError("AOI extraction runs on the GPU (tq_glimpse_extract); there is no CPU path")
    lib = _lib.load()
    H, W, F = glimpse.height, glimpse.width, glimpse.F
    N, C = images.shape[0], images.shape[2]
    frame_bytes = 2 * H * W
    nfc = max(1, min(F, (chunk_bytes or CHUNK_BYTES) // frame_bytes, (2 ** 31 - 1) // (P * P) - 1))
    frames = glimpse.cumdrift.index.values
    host = [torch.empty(nfc * frame_bytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
    free = [None, None]  # event after which a pinned buffer may be overwritten
    stream = torch.cuda.current_stream(device)
    try:
        bar = progress_bar(total=F) if progress_bar is not None else None
    except TypeError:  # a progress bar that only wraps iterables
        bar = None
    for k, f0 in enumerate(range(0, F, nfc)):
        nf = min(nfc, F - f0)
        buf = host[k % 2]
        if free[k % 2] is not None:
            free[k % 2].synchronize()
        glimpse.read_raw(frames[f0:f0 + nf], buf.numpy()[:nf * frame_bytes])
        dev = buf[:nf * frame_bytes].to(device, non_blocking=True)
        xy = torch.from_numpy(np.ascontiguousarray(raw_xy[:, f0:f0 + nf])).to(device)
        for n0 in range(0, N, 65535):  # grid.y limit of one launch
            nn = min(65535, N - n0)
            a = _lib.GlimpseArgs(
                frames=dev.data_ptr(), raw_xy=xy[n0:].data_ptr(), images=images[n0:].data_ptr(),
                target_xy=target_xy[n0:].data_ptr(), offset_hist=hist.data_ptr() if n0 == 0 else None,
                status=status.data_ptr(), H=H, W=W, N=nn, F=F, C=C, P=P, c=c, f0=f0, nf=nf,
                offset_x=int(glimpse.offset_x), offset_y=int(glimpse.offset_y), offset_P=int(offset_P))
            _lib.check(lib.tq_glimpse_extract(a, stream.cuda_stream), "tq_glimpse_extract")
        free[k % 2] = torch.cuda.Event()
        free[k % 2].record(stream)
        if bar is not None:
            bar.update(nf)
    if bar is not None:
        bar.close()


def read_glimpse(path, progress_bar=None, device="cuda", **kwargs):
    """
    Preprocess glimpse files (glimpse_reader.py:304-470): same keyword arguments (the ``.tapqir/config.yaml`` keys
    ``P``, ``num-channels``, ``dataset``, ``channels``, ``offset-P``, ``offset-x``, ``offset-y``, ``bin-size``,
    ``frame-range``, ``frame-start``, ``frame-end``, ``use-offtarget``, ``labels``), same ``data.tpqr``.  Returns the
    CosmosDataset it saved.
    """
    kwargs = dict(kwargs)
    kwargs.pop("cd", None)
    P, C = kwargs.pop("P"), kwargs.pop("num-channels")
    name, channels = kwargs.pop("dataset"), kwargs.pop("channels")
    offset_P, bin_size = kwargs.pop("offset-P"), kwargs.pop("bin-size")
    device = torch.device(device)
    if device.type != "cuda" or not torch.cuda.is_available():
        raise _lib.HipExtensionError("read_glimpse extracts the AOIs on the GPU (tq_glimpse_extract); there is no CPU path")
    _lib.load()

    images = target_xy = hist = status = None
    labels = defaultdict(list)
    time1, ttb = [], []
    for c in range(C):
        logger.info(f"Channel #{c} ({channels[c]['name']})")
        glimpse = GlimpseDataset(**kwargs, **channels[c], c=c)
        time1.append(float(glimpse.header["time1"]))
        ttb.append(glimpse.cumdrift["ttb"].values)
        drift = glimpse.cumdrift[["dx", "dy"]].values
        raw_xy = np.concatenate([np.expand_dims(glimpse.aoiinfo[d][["x", "y"]].values, axis=1) + drift for d in glimpse.dtypes], 0)
        for d in glimpse.dtypes:
            labels[d].append(glimpse.labels[d])
        if images is None:
            counts = [len(glimpse.aoiinfo[d]) for d in glimpse.dtypes]
            Nt, F = sum(counts), glimpse.F
            images = torch.zeros(Nt, F, C, P, P, dtype=torch.int32, device=device)
            target_xy = torch.zeros(Nt, F, C, 2, dtype=torch.float64, device=device)
            hist = torch.zeros(65536, dtype=torch.int64, device=device)
            status = torch.tensor([0, torch.iinfo(torch.int32).max], dtype=torch.int32, device=device)
            is_ontarget = torch.cat([torch.full((n,), d == "ontarget", dtype=torch.bool) for n, d in zip(counts, glimpse.dtypes)])
        elif raw_xy.shape[:2] != (images.shape[0], images.shape[1]):
            raise ValueError(f"channel {c} has {raw_xy.shape[0]} AOIs x {raw_xy.shape[1]} frames, "
                             f"channel 0 has {images.shape[0]} x {images.shape[1]}")
        extract_aois(glimpse, raw_xy, P, images, target_xy, hist, status, c, offset_P, progress_bar)
        outside, _ = status.tolist()
        if outside:  # the reference stops here with numpy's "could not broadcast" ValueError (glimpse_reader.py:376-378)
            raise ValueError(f"channel {c}: {outside} AOI windows of {P}x{P} pixels leave the {glimpse.height}x{glimpse.width} "
                             "frame after drift correction")
        # target positions lie within the central pixel (glimpse_reader.py:386-389)
        assert bool((target_xy[:, :, c] > 0.5 * P - 1).all()) and bool((target_xy[:, :, c] < 0.5 * P).all())

    logger.info("Processing extracted AOIs ...")
    min_data = int(status[1].item())
    offset_samples, offset_weights = _finish_offsets(hist.cpu().numpy(), min_data, bin_size)
    label_parts = []
    for d in glimpse.dtypes:
        if not any(lab is None for lab in labels[d]):
            label_parts.append(np.stack(labels[d], -1))
    dataset = CosmosDataset(
        images.cpu().long(), target_xy.cpu(), is_ontarget,
        labels=np.concatenate(label_parts, 0) if label_parts else None,
        offset_samples=offset_samples, offset_weights=offset_weights,
        time1=torch.as_tensor(time1), ttb=torch.as_tensor(np.array(ttb)).T,
        name=name, channels=tuple(ch["name"] for ch in channels))
    logger.info(f"Dataset: N={dataset.N} on-target AOIs, Nc={dataset.Nc} off-target AOIs, F={dataset.F} frames, "
                f"C={dataset.C} channels, Px={dataset.P} pixels, Py={dataset.P} pixels")
    save(dataset, path)
    return dataset
