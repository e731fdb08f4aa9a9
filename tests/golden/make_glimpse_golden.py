"""Writes tests/golden/glimpse_golden.npz: the oracle's output (oracle/glimpse.py) for the seeded synthetic experiment
of tests/glimpse_fixture.py, so that later changes of the oracle or of the fixture writer show up as a diff of data.

    python tests/golden/make_glimpse_golden.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "..", ".."), os.path.join(HERE, "..")]
from glimpse_fixture import write_experiment  # noqa: E402
from oracle import glimpse as og  # noqa: E402

CASE = dict(seed=3, C=2, P=9, F=12, n_on=5, n_off=3, labels=True)


def build():
    with tempfile.TemporaryDirectory() as td:
        cfg, _ = write_experiment(td, **CASE)
        cfg["bin-size"] = 3
        out = og.read_glimpse(**cfg)
    return dict(images=out["images"].numpy().astype(np.int32), xy=out["xy"].numpy(),
                is_ontarget=out["is_ontarget"].numpy(), offset_samples=out["offset_samples"].numpy(),
                offset_weights=out["offset_weights"].numpy(), labels_z=out["labels"]["z"], ttb=out["ttb"].numpy())


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "glimpse_golden.npz"), **build())
    print("written", os.path.getsize(os.path.join(HERE, "glimpse_golden.npz")), "bytes")
