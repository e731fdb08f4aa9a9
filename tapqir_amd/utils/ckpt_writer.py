"""
Checkpoint files of a running fit, written by a helper PROCESS.

``Model.run`` checkpoints every 200 iterations (tapqir/models/model.py:205-219, 239-289).  At the reference's default
minibatch this framework makes 200 iterations in 13 ms, while ``torch.save`` of a c2-sized state (parameters + both Adam
moments, 86 MB) takes ~30 ms and holds the interpreter lock for most of it: written in-process -- even from a thread --
the files cost more than the fit.  So inside ``run()``:

* the state is snapshotted on the DEVICE (three device-to-device copies, ~50 us), brought to a host buffer by a copy on a
  side stream that a helper thread issues and waits for (``copy_`` and the waits drop the interpreter lock), and
  written by a child process that maps the same host buffer (a file in /dev/shm, page-locked in the parent when the
  runtime allows it);
* one file is in flight at a time: a checkpoint that finds the writer busy skips its FILE (the convergence bookkeeping
  of the checkpoint is not skipped) and ``run()`` writes the final state when it ends, so the file on disk is never
  older than it would have been, only written less often.

The child never touches the GPU.  File format and payload are those of the in-process path (``Model._param_store_state``
/ ``_optim_state``): same keys, tensors as views of one storage.

Child protocol (stdin/stdout, binary): ``ready\n`` once at start, then per file 8-byte little-endian length + pickled manifest -> ``ok\n`` or ``error: ...\n``.
The manifest is produced by this module in the parent process (not a file from elsewhere).
"""

import os
import pickle
import struct
import subprocess
import sys
import threading
from pathlib import Path

import torch

_COUNTER = [0]


def build_payload(host, m):
    """The checkpoint dict from the flat host buffer ``[params | exp_avg | exp_avg_sq]`` and the manifest ``m``."""
    n = m["n"]

    def views(base):
        out = {}
        for name, (off, shape) in m["slots"].items():
            k = 1
            for s in shape:
                k *= s
            out[name] = host[base + off:base + off + k].view(shape)
        return out

    p, e1, e2 = views(0), views(n), views(2 * n)
    adam = m["adam"]
    optim = {}
    for name in p:
        optim[name] = {
            "state": {0: {"step": torch.tensor(float(adam["step"])), "exp_avg": e1[name], "exp_avg_sq": e2[name]}},
            "param_groups": [{"lr": adam["lr"], "betas": tuple(adam["betas"]), "eps": adam["eps"], "weight_decay": 0,
                              "amsgrad": False, "maximize": False, "params": [0]}],
        }
    return {
        "iter": m["iter"],
        "params": {"params": p, "constraints": {k: m["constraints"][k] for k in p}},
        "optimizer": optim,
        "rolling": m["rolling"],
        "convergence_status": m["convergence_status"],
    }


def write_file(payload, target):
    target = Path(target)
    tmp = target.with_suffix(target.suffix + ".tmp")
    torch.save(payload, tmp)
    tmp.replace(target)


class CheckpointWriter:
    """Owns the staging buffers and the child process; see the module docstring."""

    def __init__(self, nfloats, device):
        self.n = int(nfloats)
        self.device = torch.device(device)
        _COUNTER[0] += 1
        self.path = f"/dev/shm/tapqir_amd_ckpt_{os.getpid()}_{_COUNTER[0]}"
        total = 3 * self.n
        with open(self.path, "wb") as f:
            f.truncate(4 * total)
        self.host = torch.from_file(self.path, shared=True, size=total, dtype=torch.float32)
        self.pinned = False
        self.stream = None
        self.snapshot = None
        if self.device.type == "cuda":
            self.snapshot = torch.empty(total, dtype=torch.float32, device=self.device)
            self.stream = torch.cuda.Stream(device=self.device)
            try:  # page-lock the mapping so the device-to-host copy is a DMA that does not block the host
                rc = torch.cuda.cudart().cudaHostRegister(self.host.data_ptr(), 4 * total, 0)
                self.pinned = int(rc) == 0
            except Exception:
                self.pinned = False
        env = dict(os.environ)
        root = str(Path(__file__).resolve().parents[2])
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        self.child = subprocess.Popen([sys.executable, "-m", "tapqir_amd.utils.ckpt_writer", self.path, str(total)],
                                      stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env, bufsize=0)
        self._ready = False
        self._thread = None
        self._error = None
        self._closed = False
        self.failure = None  # why the child is gone (torch import failure, missing /dev/shm file, OOM kill, ...)
        self.files_written = 0
        import atexit
        import weakref

        ref = weakref.ref(self)
        atexit.register(lambda: ref() is not None and ref().close())

    # ---------------------------------------------------------------------------------------------------
    def ready(self):
        """True once the child has started (it needs ~1 s to import torch); never blocks."""
        if not self._ready and self._thread is None:
            import select

            if select.select([self.child.stdout], [], [], 0)[0]:
                self._set_ready(self.child.stdout.readline())
        return self._ready

    def failed(self):
        """True once the child process is known to be gone (it never restarts): the caller drops this writer and writes
        its files in-process.  Never blocks."""
        if self.failure is None and not self._closed and self.child.poll() is not None:
            self.failure = f"exit code {self.child.returncode}"
        return self.failure is not None

    def _set_ready(self, line):
        self._ready = line.strip() == b"ready"
        if not self._ready:  # EOF (an empty read) or anything but the greeting: the child did not come up
            self.failure = self.failure or ("ended before it was ready" if not line else f"unexpected greeting {line[:40]!r}")
        if self._ready:  # both processes have the buffer mapped: the name can go (nothing is left behind on a crash)
            try:
                os.unlink(self.path)
            except OSError:
                pass

    def busy(self):
        t = self._thread
        return t is not None and t.is_alive()

    def join(self):
        t = self._thread
        if t is not None:
            t.join()
            self._thread = None
        if self._error is not None:
            err, self._error = self._error, None
            raise RuntimeError(f"checkpoint writer: {err}")

    def submit(self, params, exp_avg, exp_avg_sq, manifest, target):
        """Snapshot the three flat buffers now (in stream order) and write ``target`` in the background."""
        self.join()
        n = self.n
        blob = pickle.dumps(dict(manifest, n=n, target=str(target)))
        taken = None
        if self.device.type == "cuda":
            snap = self.snapshot
            snap[:n].copy_(params)
            snap[n:2 * n].copy_(exp_avg)
            snap[2 * n:].copy_(exp_avg_sq)
            taken = torch.cuda.Event()
            taken.record()
        else:
            self.host[:n].copy_(params)
            self.host[n:2 * n].copy_(exp_avg)
            self.host[2 * n:].copy_(exp_avg_sq)

        def run():
            try:
                if taken is not None:  # device -> host on the side stream, from this thread (copy_ drops the GIL)
                    with torch.cuda.stream(self.stream):
                        self.stream.wait_event(taken)
                        self.host.copy_(self.snapshot, non_blocking=self.pinned)
                        self.stream.synchronize()
                if not self._ready:  # (a caller that did not poll ready(): wait for the child to come up)
                    self._set_ready(self.child.stdout.readline())
                self.child.stdin.write(struct.pack("<Q", len(blob)) + blob)
                reply = self.child.stdout.readline().decode().strip()
                if reply != "ok":
                    self._error = reply or "the writer process ended"
                    if not reply:
                        self.failure = self.failure or "ended while writing a file"
                else:
                    self.files_written += 1
            except Exception as err:  # (BrokenPipeError when the child is gone)
                self._error = repr(err)
                if isinstance(err, OSError):
                    self.failure = self.failure or repr(err)

        self._thread = threading.Thread(target=run, name="tapqir-checkpoint")
        self._thread.start()

    def close(self):
        """Wait for the file in flight, end the child (without waiting for its interpreter to unwind), free the buffer."""
        if self._closed:
            return
        self._closed = True
        try:
            self.join()
        finally:
            try:
                self.child.stdin.close()
            except Exception:  # pragma: no cover
                pass
            if self.pinned:
                try:
                    torch.cuda.cudart().cudaHostUnregister(self.host.data_ptr())
                except Exception:  # pragma: no cover
                    pass
            try:
                os.unlink(self.path)
            except OSError:
                pass

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def _child_main(path, total):
    host = torch.from_file(path, shared=True, size=total, dtype=torch.float32)
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    try:
        out.write(b"ready\n")
        out.flush()
    except BrokenPipeError:  # the parent is gone already
        os._exit(0)
    while True:
        head = inp.read(8)
        if len(head) < 8:
            out.flush()
            os._exit(0)  # the parent closed the pipe: nothing to unwind
        (size,) = struct.unpack("<Q", head)
        manifest = pickle.loads(inp.read(size))
        try:
            write_file(build_payload(host, manifest), manifest["target"])
            out.write(b"ok\n")
        except Exception as err:
            out.write(("error: " + repr(err).replace("\n", " ") + "\n").encode())
        out.flush()


if __name__ == "__main__":
    _child_main(sys.argv[1], int(sys.argv[2]))
