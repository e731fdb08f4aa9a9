"""Writes a synthetic Glimpse / imscroll experiment folder (header.mat, N.glimpse, driftlist, aoiinfo, intervals) in
the formats tapqir/imscroll/glimpse_reader.py:57-186 reads, and returns the config dict ``read_glimpse`` takes."""

import os

import numpy as np
from scipy.io import savemat


def analytic_frame(f, H, W):
    """Pixel value that encodes its own position: checks of the crop need no second implementation."""
    r, c = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    return ((r * 977 + c * 31 + f * 7919) % 65536).astype(np.int64)


def noisy_frame(rng, H, W, hot=True):
    """Camera-like counts: offset ~ 90 +- a few ADU everywhere, a bright band, extremes and (optionally) hot pixels
    far above the offset peak (outside the LDS window of the histogram kernel)."""
    img = rng.integers(80, 101, size=(H, W)).astype(np.int64)
    img[H // 2:] += rng.integers(0, 800, size=(H - H // 2, W))
    img[-1, -1], img[-1, 0] = 65535, 0
    if hot:
        img[12, 12] = 40000 + rng.integers(0, 3)
        img[13, 17] = 65535
    return img


def write_experiment(root, H=64, W=72, F=12, n_on=5, n_off=3, C=1, P=14, seed=0, kind="noisy", labels=True,
                     frame_range=None, aoiinfo_frame=4, drift_scale=0.3):
    """-> (config kwargs for read_glimpse, list over channels of (F_all, H, W) int arrays of the true pixel values)."""
    rng = np.random.default_rng(seed)
    os.makedirs(root, exist_ok=True)
    channels, truth = [], []
    for c in range(C):
        folder = os.path.join(root, f"ch{c}")
        os.makedirs(folder, exist_ok=True)
        frames = np.stack([analytic_frame(f + 100 * c, H, W) if kind == "analytic" else noisy_frame(rng, H, W)
                           for f in range(F)])
        truth.append(frames)
        # two .glimpse files; the second one has a gap of junk bytes between frames (non-contiguous offsets)
        split = F // 2
        filenumber, offset = np.zeros(F, dtype=np.int64), np.zeros(F, dtype=np.int64)
        with open(os.path.join(folder, "0.glimpse"), "wb") as fa, open(os.path.join(folder, "1.glimpse"), "wb") as fb:
            for f in range(F):
                fid = fa if f < split else fb
                if f >= split and f % 2 == 1:
                    fid.write(b"\xab" * 10)
                filenumber[f], offset[f] = (0 if f < split else 1), fid.tell()
                fid.write((frames[f] - 2 ** 15).astype(">i2").tobytes())
        savemat(os.path.join(folder, "header.mat"), {"vid": {
            "height": float(H), "width": float(W), "nframes": float(F), "filenumber": filenumber.astype(np.uint8),
            "offset": offset.astype(np.uint32), "ttb": 1000.0 * np.arange(F) + 17 * c, "time1": 3.7e9 + c}})

        drift = np.zeros((F, 4))
        drift[:, 0] = np.arange(1, F + 1)
        drift[:, 1:3] = rng.normal(0, drift_scale, size=(F, 2))
        drift[aoiinfo_frame - 1, 1:3] = 0.0 if c == 0 else [0.125, -0.25]  # the aoiinfo frame keeps its raw entry
        savemat(os.path.join(folder, "driftlist.mat"), {"driftlist": drift})

        def table(n, first_aoi):
            t = np.zeros((n, 6))
            t[:, 0] = aoiinfo_frame
            t[:, 1] = 1
            t[:, 2] = rng.uniform(P + 2, H - P - 2, n)  # y (1-based)
            t[:, 3] = rng.uniform(P + 2, W - P - 2, n)  # x
            t[:, 4] = P
            t[:, 5] = np.arange(first_aoi, first_aoi + n)
            return t

        on, off = table(n_on, 1), table(n_off, 1)
        savemat(os.path.join(folder, "on.mat"), {"aoiinfo2": on})
        if c == 0:
            np.savetxt(os.path.join(folder, "off.dat"), off)  # plain-text aoiinfo (glimpse_reader.py:82-83)
            off_path = os.path.join(folder, "off.dat")
        else:
            savemat(os.path.join(folder, "off.mat"), {"aoifits": {"aoiinfo2": off}})
            off_path = os.path.join(folder, "off.mat")
        ch = {"name": f"dye{c}", "glimpse-folder": folder, "driftlist": os.path.join(folder, "driftlist.mat"),
              "ontarget-aoiinfo": os.path.join(folder, "on.mat"), "offtarget-aoiinfo": off_path}
        if labels:
            cia = np.array([[-2, 1, 3, 3, 0, 0, 1], [1, 4, 7, 4, 0, 0, 1], [0, 8, F, F - 7, 0, 0, 1],
                            [-3, 1, 5, 5, 0, 0, 2], [2, 6, F, F - 5, 0, 0, 2], [3, 2, 2, 1, 0, 0, 4]], dtype=float)
            savemat(os.path.join(folder, "labels.mat"), {"Intervals": {"CumulativeIntervalArray": cia}})
            ch["ontarget-labels"] = os.path.join(folder, "labels.mat")
            ch["offtarget-labels"] = None
        channels.append(ch)
    cfg = {"P": P, "num-channels": C, "dataset": "synthetic", "channels": channels, "offset-P": 20, "offset-x": 4,
           "offset-y": 6, "bin-size": 1, "frame-range": frame_range is not None,
           "frame-start": frame_range[0] if frame_range else None, "frame-end": frame_range[1] if frame_range else None,
           "use-offtarget": n_off > 0, "labels": labels}
    return cfg, truth


def write_frames_experiment(root, frames, on_xy, off_xy, drift_dxdy, aoiinfo_frame, P, offset_x, offset_y, offset_P,
                            name="sim"):
    """One channel from explicit arrays: ``frames`` (F, H, W) counts 0..65535, AOI positions (0-based x, y) at the
    aoiinfo frame, per-frame drift increments (F, 2) as (dx, dy) -> config dict for read_glimpse."""
    os.makedirs(root, exist_ok=True)
    F, H, W = frames.shape
    with open(os.path.join(root, "0.glimpse"), "wb") as fid:
        fid.write((frames - 2 ** 15).astype(">i2").tobytes())
    savemat(os.path.join(root, "header.mat"), {"vid": {
        "height": float(H), "width": float(W), "nframes": float(F), "filenumber": np.zeros(F, dtype=np.uint8),
        "offset": (np.arange(F, dtype=np.uint64) * 2 * H * W).astype(np.uint32), "ttb": 100.0 * np.arange(F), "time1": 1.0}})
    drift = np.zeros((F, 3))
    drift[:, 0] = np.arange(1, F + 1)
    drift[:, 1], drift[:, 2] = drift_dxdy[:, 1], drift_dxdy[:, 0]  # columns: frame, dy, dx
    savemat(os.path.join(root, "driftlist.mat"), {"driftlist": drift})

    def table(xy):
        t = np.zeros((len(xy), 6))
        t[:, 0], t[:, 1], t[:, 4] = aoiinfo_frame, 1, P
        t[:, 3], t[:, 2] = xy[:, 0] + 1, xy[:, 1] + 1  # MATLAB coordinates
        t[:, 5] = np.arange(1, len(xy) + 1)
        return t

    savemat(os.path.join(root, "on.mat"), {"aoiinfo2": table(on_xy)})
    savemat(os.path.join(root, "off.mat"), {"aoiinfo2": table(off_xy)})
    ch = {"name": name, "glimpse-folder": root, "driftlist": os.path.join(root, "driftlist.mat"),
          "ontarget-aoiinfo": os.path.join(root, "on.mat"), "offtarget-aoiinfo": os.path.join(root, "off.mat")}
    return {"P": P, "num-channels": 1, "dataset": name, "channels": [ch], "offset-P": offset_P, "offset-x": offset_x,
            "offset-y": offset_y, "bin-size": 1, "frame-range": False, "frame-start": None, "frame-end": None,
            "use-offtarget": True, "labels": False}
