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


if __name__ == "__main__":
    main()
