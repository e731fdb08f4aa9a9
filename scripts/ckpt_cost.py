"""Where a checkpoint of a c2-sized fit spends its time (Model.save_checkpoint every 200 iterations)."""
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
    m.run(10, progress_bar=lambda r: r)
    m.iter_loss = 0.0
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.engine.join(); ok = bool(torch.isfinite(m.engine.params).all()); torch.cuda.synchronize(); t1 = time.perf_counter()
        ps = m._param_store_state(); os_ = m._optim_state(); t2 = time.perf_counter()
        torch.save({"iter": 1, "params": ps, "optimizer": os_, "rolling": {}, "convergence_status": False}, os.path.join(td, "x.tpqr")); t3 = time.perf_counter()
        print(f"join+isfinite {1e3*(t1-t0):.1f} ms, device->host + views {1e3*(t2-t1):.1f} ms, torch.save {1e3*(t3-t2):.1f} ms")
    t0 = time.perf_counter(); m.save_checkpoint(); print(f"save_checkpoint total {1e3*(time.perf_counter()-t0):.1f} ms")
