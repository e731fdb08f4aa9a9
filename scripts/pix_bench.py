"""Time only the fused render + log-prob kernel at a given dataset size (HIP events); used under rocprofv3 --pmc."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--aois", type=int, default=400)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--K", type=int, default=2)
    ap.add_argument("--P", type=int, default=14)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--offsets", default="sim")
    a = ap.parse_args()
    from tapqir_amd.models.cosmos import initial_values
    from tapqir_amd.models.engine import CosmosEngine
    from tapqir_amd.utils.dataset import CosmosDataset
    from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

    dev = torch.device("cuda", 0)

    class _M:
        K, device = a.K, dev

    data = simulate(_M, a.aois, a.frames, 1, a.P, seed=1000, params=TEST_PARAMS)
    if a.offsets == "hist":
        s = torch.arange(70.0, 120.0)
        w = torch.minimum(s - 69.0, 120.0 - s)
        data = CosmosDataset(data.images, data.xy, data.is_ontarget, offset_samples=s, offset_weights=w / w.sum())
    eng = CosmosEngine(data, K=a.K, device=dev, seed=7)
    eng.layout.set_constrained(eng.params, initial_values(eng, data))
    B = a.aois * a.frames
    for bwd in (False, True):
        t = bench.time_pixel_kernel(eng, a.launches, bwd)
        by = bench.fwd_bytes_per_unit(a.K, a.P) * B
        print(f"pixel kernel K={a.K} P={a.P} units={B} bwd={int(bwd)}: {t*1e6:.1f} us  {by/t/1e9:.0f} GB/s algorithmic", flush=True)


if __name__ == "__main__":
    main()
