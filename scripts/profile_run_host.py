"""Host-side profile (cProfile) of Model.run at the default minibatch: where the ~60 us of Python + launch per iteration go."""
import cProfile, os, pstats, sys, tempfile
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tapqir_amd.models import models
from tapqir_amd.utils.dataset import save
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate
class _M: K, device = 2, torch.device("cuda", 0)
with tempfile.TemporaryDirectory() as td:
    save(simulate(_M, 400, 1000, 1, 14, seed=2, params=TEST_PARAMS), td)
    m = models["cosmos"](S=1, K=2, device="cuda", dtype="double")
    m.load(td); m.init(lr=0.005, nbatch_size=10, fbatch_size=512); m.run_path = None
    m.run(600, progress_bar=lambda r: r)
    pr = cProfile.Profile(); pr.enable()
    m.run(4000, progress_bar=lambda r: r)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(16)
