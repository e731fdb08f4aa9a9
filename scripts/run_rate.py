"""Iterations per second of Model.run at the reference's default minibatch (10 AOIs x 512 frames) on the c2 data set."""
import os, sys, time, tempfile
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tapqir_amd.models import models
from tapqir_amd.utils.dataset import save
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

class _M: K, device = 2, torch.device("cuda", 0)
with tempfile.TemporaryDirectory() as td:
    save(simulate(_M, 400, 1000, 1, 14, seed=2, params=TEST_PARAMS), td)
    m = models["cosmos"](S=1, K=2, device="cuda", dtype="double")
    m.load(td)
    m.init(lr=0.005, nbatch_size=10, fbatch_size=512)
    if os.environ.get("RATE_NO_CKPT"):
        m.run_path = None  # no checkpoint file (convergence bookkeeping only)
    m.run(400, progress_bar=lambda r: r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = int(os.environ.get("RATE_ITERS", 4000))
    m.run(N, progress_bar=lambda r: r)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    NOTE = ", no file" if os.environ.get("RATE_NO_CKPT") else ""
    print(f"Model.run 10x512 on 400x1000: {N / dt:.0f} it/s over {N} iterations ({dt / N * 1e6:.1f} us per iteration, checkpoint every 200{NOTE})")
