"""Probe: the single-workgroup tail of step t-1 and the global draws of step t on a SIDE stream, concurrent with the local
sampling of step t as the plain kernel (91 registers, five waves per SIMD) -- against the shipped step, whose sampling
launch carries the tail as one extra workgroup and is therefore compiled for three waves per SIMD."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tapqir_amd import _lib
from tapqir_amd.models.cosmos import initial_values
from tapqir_amd.models.engine import CosmosEngine
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate
dev = torch.device("cuda", 0)
class _M: K, device = 2, dev
data = simulate(_M, 400, 1000, 1, 14, seed=1000, params=TEST_PARAMS)

def make():
    e = CosmosEngine(data, K=2, device=dev, seed=7)
    e.layout.set_constrained(e.params, initial_values(e, data))
    e.pixel_mode, e.fuse_unit = 0, True
    return e

def shipped(n):
    e = make()
    for _ in range(30): e.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): e.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e6
    e.join(); return dt, float(e.elbo_out[0])

def two_streams(n):
    e = make()
    M = torch.cuda.current_stream(dev)
    S = torch.cuda.Stream(dev)
    hM, hS = C.c_void_p(M.cuda_stream), C.c_void_p(S.cuda_stream)
    lib = e.lib
    prev = None
    def step():
        nonlocal prev
        a = e._step_args(None, None)
        a.fuse_adam, a.last_step, a.pixel_mode = 1, None, 2
        evA, evB = torch.cuda.Event(), torch.cuda.Event()
        evA.record(M)
        S.wait_event(evA)
        if prev is not None:
            _lib.check(lib.tq_cosmos_tail(C.byref(prev), hS), "tail")
        _lib.check(lib.tq_cosmos_sample_globals(C.byref(a), hS), "sample_globals")
        evB.record(S)
        _lib.check(lib.tq_cosmos_sample_locals(C.byref(a), hM), "sample_locals")
        M.wait_event(evB)
        _lib.check(lib.tq_cosmos_pixel_unit(C.byref(a), hM), "pixel_unit")
        prev = a
        e.adam_step += 1
    for _ in range(30): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e6
    _lib.check(lib.tq_cosmos_tail(C.byref(prev), hM), "tail"); torch.cuda.synchronize()
    return dt, float(e.elbo_out[0])

for rep in range(2):
    a, ea = shipped(200)
    b, eb = two_streams(200)
    print(f"shipped {a:.1f} us/step (ELBO {ea:.6g}), tail on a side stream {b:.1f} us/step (ELBO {eb:.6g})")
