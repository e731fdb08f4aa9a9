"""Wall clock of `tapqir_amd fit` at the reference's default minibatch (10 AOIs x 512 frames) on simulated data."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ["PYTHONPATH"] = ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")

from tapqir_amd.utils.dataset import save
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

N, F, iters = 100, 1000, int(sys.argv[1]) if len(sys.argv) > 1 else 4000
with tempfile.TemporaryDirectory() as td:
    save(simulate(2, N, F, 1, 14, seed=2, params=TEST_PARAMS), td)
    t0 = time.perf_counter()
    rc = subprocess.call([sys.executable, "-m", "tapqir_amd", "--cd", td, "fit", "--model", "cosmos", "--cuda", "--nbatch-size", "10",
                          "--fbatch-size", "512", "--learning-rate", "0.005", "--num-iter", str(iters), "--no-input"],
                         stdout=subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    import pandas as pd

    summ = pd.read_csv(f"{td}/cosmos_summary.csv", index_col=0)
    print(f"rc={rc} {iters} iterations of 10x512 on {N}x{F}: {dt:.2f} s wall (process start, load, fit, stats)")
    print(summ.loc[[i for i in ("MCC", "Recall", "Precision", "gain", "proximity", "lamda", "SNR") if i in summ.index]].to_string())
