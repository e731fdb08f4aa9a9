"""Where do the ~36 us of the sharded step path go?  One rank, full-batch steps with (a) no all-reduce (single-GPU
pipeline), (b) a dummy handle (sharded launch sequence, no process group), (c) torch's RCCL all-reduce, (d) the same
issued from a side stream."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tapqir_amd.models.cosmos import initial_values
from tapqir_amd.models.engine import CosmosEngine
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)


class _M:
    K, device = 2, dev


data = simulate(_M, 400, 1000, 1, 14, seed=1000, params=TEST_PARAMS)


class Dummy:
    def wait(self):
        pass


side = torch.cuda.Stream(dev)


def side_allreduce(t):
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        h = dist.all_reduce(t, async_op=True)

    class H:
        def wait(self_inner):
            h.wait()  # side stream waits for the collective
            torch.cuda.current_stream(dev).wait_stream(side)

    return H()


for name, ar in (("single", None), ("dummy", lambda t: Dummy()), ("rccl", lambda t: dist.all_reduce(t, async_op=True)),
                 ("rccl-side", side_allreduce)):
    eng = CosmosEngine(data, K=2, device=dev, seed=7)
    eng.layout.set_constrained(eng.params, initial_values(eng, data))
    for _ in range(300):
        eng.step(allreduce=ar)
    eng.join()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        eng.step(allreduce=ar)
    eng.join()
    torch.cuda.synchronize()
    print(f"{name:10s} {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms/step", flush=True)
dist.destroy_process_group()
