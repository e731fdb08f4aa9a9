"""Throughput of tq_glimpse_extract (AOI extraction from raw frames) with the frames already in HBM, against its
algorithmic bytes: per AOI-frame 2 P^2 (big-endian int16 gathered) + 4 P^2 (int32 written) + 16 + 16 (positions).

    python scripts/glimpse_bench.py [--H 512 --W 512 --frames 200 --aois 800 --P 14]
"""

import argparse
import json

import numpy as np
import torch

from tapqir_amd import _lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=512)
    ap.add_argument("--W", type=int, default=512)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--aois", type=int, default=800)
    ap.add_argument("--P", type=int, default=14)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda")
    H, W, F, N, P = a.H, a.W, a.frames, a.aois, a.P
    rng = np.random.default_rng(0)
    frames = torch.from_numpy(rng.integers(0, 256, size=F * H * W * 2, dtype=np.uint8)).to(dev)
    xy = np.stack([rng.uniform(P, W - P, (N, 1)), rng.uniform(P, H - P, (N, 1))], -1) + rng.normal(0, 0.5, (1, F, 2)).cumsum(1) * 0.1
    raw = torch.from_numpy(np.ascontiguousarray(xy)).to(dev)
    images = torch.zeros(N, F, 1, P, P, dtype=torch.int32, device=dev)
    txy = torch.zeros(N, F, 1, 2, dtype=torch.float64, device=dev)
    hist = torch.zeros(65536, dtype=torch.int64, device=dev)
    status = torch.tensor([0, 2 ** 31 - 1], dtype=torch.int32, device=dev)
    args = _lib.GlimpseArgs(frames=frames.data_ptr(), raw_xy=raw.data_ptr(), images=images.data_ptr(), target_xy=txy.data_ptr(),
                            offset_hist=hist.data_ptr(), status=status.data_ptr(), H=H, W=W, N=N, F=F, C=1, P=P, c=0, f0=0, nf=F,
                            offset_x=10, offset_y=10, offset_P=30)
    st = torch.cuda.current_stream().cuda_stream
    def timed():
        for _ in range(3):
            _lib.check(lib.tq_glimpse_extract(args, st), "tq_glimpse_extract")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            lib.tq_glimpse_extract(args, st)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    ms = timed()
    args.offset_hist = None
    ms_crop = timed()
    units = N * F
    algo = units * (6 * P * P + 32)
    print(json.dumps({"kernel": "tq_glimpse_extract", "H": H, "W": W, "frames": F, "aois": N, "P": P, "ms": ms, "ms_crop_only": ms_crop, "crop_GBps": N * F * (6 * P * P + 32) / ms_crop / 1e6,
                      "aoi_frames_per_s": units / ms * 1e3, "algorithmic_GBps": algo / ms / 1e6,
                      "frac_of_8TBps": algo / ms / 1e6 / 8000, "outside": status[0].item()}))




def end_to_end(H=512, W=512, F=300, n_on=400, n_off=200, P=14):
    """Wall clock of read_glimpse (files -> data.tpqr) on a synthetic experiment, next to the oracle's frame loop on a
    slice of it."""
    import os
    import sys
    import tempfile
    import time

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    from glimpse_fixture import write_experiment
    from oracle import glimpse as og
    from tapqir_amd.imscroll import read_glimpse

    with tempfile.TemporaryDirectory() as td:
        cfg, _ = write_experiment(os.path.join(td, "raw"), H=H, W=W, F=F, n_on=n_on, n_off=n_off, P=P, kind="noisy", labels=False,
                                  drift_scale=0.05, aoiinfo_frame=F // 2)
        read_glimpse(td, None, **cfg)  # warm-up: library load, pinned buffers
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ds = read_glimpse(td, None, **cfg)
        torch.cuda.synchronize()
        t_gpu = time.perf_counter() - t0
        cfg_small = dict(cfg, **{"frame-range": True, "frame-start": F // 2 - 4, "frame-end": F // 2 + 5})
        t0 = time.perf_counter()
        og.read_glimpse(**cfg_small)
        t_cpu = (time.perf_counter() - t0) * F / 10
    units = (n_on + n_off) * F
    print(json.dumps({"read_glimpse_s": t_gpu, "aoi_frames": units, "aoi_frames_per_s": units / t_gpu,
                      "frame_MB": F * H * W * 2 / 1e6, "oracle_loop_s_extrapolated": t_cpu, "speedup": t_cpu / t_gpu,
                      "images_shape": list(ds.images.shape)}))


if __name__ == "__main__":
    import sys as _sys

    if "--end-to-end" in _sys.argv:
        end_to_end()
    else:
        main()
