"""cProfile of Model.run at the default minibatch (host side of the fit loop)."""
import cProfile
import pstats
import tempfile

from tapqir_amd.models import models
from tapqir_amd.utils.dataset import save
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

with tempfile.TemporaryDirectory() as td:
    save(simulate(2, 100, 1000, 1, 14, seed=2, params=TEST_PARAMS), td)
    m = models["cosmos"](S=1, K=2, device="cuda", dtype="double")
    m.load(td)
    m.init(lr=0.005, nbatch_size=10, fbatch_size=512)
    m.run(400, progress_bar=lambda x: x)
    pr = cProfile.Profile()
    pr.enable()
    m.run(2000, progress_bar=lambda x: x)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumtime").print_stats(22)
