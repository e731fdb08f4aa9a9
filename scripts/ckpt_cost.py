"""Where a checkpoint of a c2-sized fit spends its time (Model.save_checkpoint every 200 iterations inside run())."""
import os, sys, time, tempfile
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tapqir_amd.models import models
from tapqir_amd.utils import ckpt_writer
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
    # in-process pieces
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.engine.join(); ok = bool(torch.isfinite(m.engine.params).all()); torch.cuda.synchronize(); t1 = time.perf_counter()
        ps = m._param_store_state(); os_ = m._optim_state(); t2 = time.perf_counter()
        torch.save({"iter": 1, "params": ps, "optimizer": os_, "rolling": {}, "convergence_status": False}, os.path.join(td, "x.tpqr")); t3 = time.perf_counter()
        print(f"in-process: join+isfinite {1e3*(t1-t0):.1f} ms, device->host + views {1e3*(t2-t1):.1f} ms, torch.save {1e3*(t3-t2):.1f} ms")
    # helper process
    eng = m.engine
    w = ckpt_writer.CheckpointWriter(eng.params.numel(), eng.params.device)
    print("host buffer page-locked:", w.pinned)
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        man = m._manifest(); t1 = time.perf_counter()
        w.submit(eng.params, eng.exp_avg, eng.exp_avg_sq, man, os.path.join(td, "y.tpqr")); t2 = time.perf_counter()
        # the main thread keeps stepping while the file is written
        n = 0
        while w.busy():
            m.step_async(); n += 1
        torch.cuda.synchronize(); t3 = time.perf_counter()
        w.join()
        print(f"process: manifest {1e3*(t1-t0):.2f} ms, submit {1e3*(t2-t1):.2f} ms, file complete after {1e3*(t3-t2):.1f} ms "
              f"({n} steps meanwhile = {1e6*(t3-t2)/max(n,1):.1f} us/step)")
    w.close()
    # main-thread cost of one checkpoint inside run(): everything in save_checkpoint
    m._in_run = True
    for rep in range(3):
        for _ in range(200):
            m.step_async()
        t0 = time.perf_counter(); m.iter_loss = m.step(); t1 = time.perf_counter(); m.save_checkpoint(); t2 = time.perf_counter()
        print(f"in run(): step()+loss readback {1e3*(t1-t0):.2f} ms, save_checkpoint {1e3*(t2-t1):.2f} ms, stale={m._ckpt_file_stale}")
    m._in_run = False
    m._join_checkpoint_writer(close=True)
