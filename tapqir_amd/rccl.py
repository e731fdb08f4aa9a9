"""
The step's one collective issued straight on the launch stream: ``ncclAllReduce`` of RCCL through ctypes.

``torch.distributed.all_reduce`` runs the same RCCL call, but on the process group's own stream: every step then pays two
event hand-overs between that stream and the stream the HIP kernels are launched on (+26 us per step at one rank, DESIGN.md
section 6).  The cross-unit sums are 6 doubles; the collective is pure latency, so that overhead is most of its cost.  Here a
second communicator is created once -- ``ncclCommInitRank`` with a unique id that rank 0 generates and ``torch.distributed``
broadcasts (any backend: the process group stays in charge of everything outside the step, ``tapqir_amd.parallel.Collective``)
-- and the all-reduce of a step is one more launch in stream order between ``tq_cosmos_elbo_grads`` and the tail: no second
stream, no events, nothing for the host to wait for.
"""

import ctypes as C
import os

import torch

NCCL_UNIQUE_ID_BYTES = 128
NCCL_FLOAT64, NCCL_SUM = 8, 0  # ncclDataType_t / ncclRedOp_t (nccl.h)


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * NCCL_UNIQUE_ID_BYTES)]


def _load():
    # the copy torch itself has loaded (same file -> same library instance), else the ROCm one
    cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so", "librccl.so"]
    err = None
    for path in cands:
        try:
            lib = C.CDLL(path)
            break
        except OSError as e:  # noqa: PERF203
            err = e
    else:
        raise RuntimeError(f"librccl.so not found ({err})")
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.ncclGetErrorString.argtypes = [C.c_int]
    for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllReduce, lib.ncclCommDestroy):
        f.restype = C.c_int
    return lib


class InStream:
    """What ``RcclDirect.__call__`` returns: the collective is ordered by the stream it was issued on, there is nothing to
    wait for.  ``CosmosEngine.step`` tells it from a ``torch.distributed`` work handle by ``in_stream``."""

    in_stream = True

    def wait(self):
        return True


class RcclDirect:
    """``allreduce(gsum)`` for ``CosmosEngine.step``: in-place sum of a float64 device tensor over the ranks of ``group``,
    issued on torch's current stream of ``device``."""

    backend = "rccl-direct"

    def __init__(self, group=None, device=None):
        import torch.distributed as dist

        self.lib = _load()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        uid = _UniqueId()
        if self.rank == 0:
            self._check(self.lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        raw = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).clone()
        on_dev = dist.get_backend(group) == "nccl"
        buf = raw.to(self.device) if on_dev else raw
        dist.broadcast(buf, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        C.memmove(C.byref(uid), bytes(buf.cpu().numpy().tobytes()), NCCL_UNIQUE_ID_BYTES)
        self.comm = C.c_void_p()
        with torch.cuda.device(self.device):
            self._check(self.lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")
        self._handle = InStream()

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.lib.ncclGetErrorString(rc).decode()}")

    def __call__(self, t: torch.Tensor):
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        stream = torch._C._cuda_getCurrentRawStream(self.device.index)
        p = C.c_void_p(t.data_ptr())
        self._check(self.lib.ncclAllReduce(p, p, t.numel(), NCCL_FLOAT64, NCCL_SUM, self.comm, C.c_void_p(stream)), "ncclAllReduce")
        return self._handle

    @classmethod
    def checked(cls, group=None, device=None):
        """An instance whose all-reduce has been checked once against the process group -- every rank contributes rank + 1 and
        must read the sum of all ranks -- or None if creating it or the check failed on ANY rank (agreed through the process
        group, so that all ranks take the same path).  No N > 1 hardware has run this path yet: callers fall back to
        ``torch.distributed.all_reduce``."""
        import sys

        import torch.distributed as dist

        inst, ok = None, 1.0
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        try:
            inst = cls(group, dev)
            t = torch.full((6,), float(inst.rank + 1), dtype=torch.float64, device=dev)
            inst(t)
            torch.cuda.synchronize(dev)
            ok = float(bool((t == inst.world * (inst.world + 1) / 2).all()))
        except Exception as err:  # noqa: BLE001
            print(f"tapqir_amd.rccl: direct all-reduce unavailable ({err}); using torch.distributed.all_reduce", file=sys.stderr)
            ok = 0.0
        flag = torch.tensor([ok], device=dev if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return inst if float(flag[0]) >= 1.0 else None

    def close(self):
        if self.comm:
            torch.cuda.synchronize(self.device)
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()
