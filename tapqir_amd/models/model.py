"""
Model driver of the drop-in surface (mirrors tapqir/models/model.py:31-371: same method names,
arguments, attributes, files and error behaviour).  ``self.svi.step()`` of the reference
(model.py:212) is ``self.step()`` here, executed by the HIP library through ``CosmosEngine``.
"""

import logging
import math
import random
from collections import defaultdict, deque
from pathlib import Path
from typing import Union

import torch

from tapqir_amd.exceptions import CudaOutOfMemoryError, TapqirFileNotFoundError
from tapqir_amd.utils.dataset import load
from tapqir_amd.utils.safe_load import load_tpqr

logger = logging.getLogger(__name__)

try:  # tensorboard is optional glue (model.py:209, 284-298)
    from torch.utils.tensorboard import SummaryWriter
except Exception:  # pragma: no cover
    SummaryWriter = None


class _NullWriter:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def add_scalar(self, *a, **k):
        pass

    def add_scalars(self, *a, **k):
        pass


def _is_oom(err):
    msg = str(err.args[0]) if err.args else ""
    return msg.startswith("CUDA out of memory") or msg.startswith("HIP out of memory") or "out of memory" in msg


class Model:
    r"""
    Base class for tapqir models (model.py:31-48).  Derived models implement ``init_parameters``
    and ``step`` (the reference's ``model`` / ``guide`` / ``TraceELBO`` are Pyro programs; the
    equivalents here are descriptions of, and handles on, the fused HIP step).
    """

    name = "model"

    def __init__(self, S: int = 1, K: int = 2, Q: int = None, device: str = "cpu", dtype: str = "double",
                 priors: dict = None):
        self.S = S
        self.K = K
        self._Q = Q
        self.nbatch_size = None
        self.fbatch_size = None
        self.priors = priors
        self.n = None
        self.f = None
        self.data_path = None
        self.path = None
        self.run_path = None
        self.engine = None
        self.collective = None  # tapqir_amd.parallel.attach: blocking collectives of an AOI-sharded fit
        self.stats_path = None  # where compute_stats writes (default: self.path)
        self.to(device, dtype)

    def to(self, device: str, dtype: str = "double") -> None:
        """model.py:75-91.  ``dtype`` is accepted for API compatibility; the HIP kernels compute
        in float32 with float64 cross-unit sums (the reference's `tapqir fit` uses float64,
        main.py:428) and no global default tensor type is changed."""
        self.dtype = getattr(torch, dtype)
        self.device = torch.device(device)

    @property
    def Q(self):
        return self._Q or self.data.C

    def load(self, path: Union[str, Path], data_only: bool = True) -> None:
        """model.py:97-127."""
        self.path = Path(path)
        self.run_path = self.path / ".tapqir"
        self.data = load(self.path, self.device)
        logger.debug(f"Loaded data from {self.path / 'data.tpqr'}")
        if not data_only:
            try:
                self.params = load_tpqr(self.path / f"{self.name}_params.tpqr")
            except FileNotFoundError:
                raise TapqirFileNotFoundError("parameter", self.path / f"{self.name}_params.tpqr")
            try:
                import pandas as pd

                self.summary = pd.read_csv(self.path / f"{self.name}_summary.csv", index_col=0)
            except FileNotFoundError:
                raise TapqirFileNotFoundError("summary", self.path / f"{self.name}_summary.csv")

    # -- subclass hooks ---------------------------------------------------------------------------
    def model(self):
        raise NotImplementedError

    def guide(self):
        raise NotImplementedError

    def TraceELBO(self, jit=False):
        raise NotImplementedError

    def init_parameters(self):
        raise NotImplementedError

    def _make_engine(self):
        raise NotImplementedError

    def step(self) -> float:
        raise NotImplementedError

    # -- model.py:153-186 ---------------------------------------------------------------------------
    def init(self, lr: float = 0.005, nbatch_size: int = 5, fbatch_size: int = 512, jit: bool = False) -> None:
        self.lr = lr
        self.optim_args = {"lr": lr, "betas": [0.9, 0.999]}
        self._make_engine()
        self.engine.lr, self.engine.betas = lr, (0.9, 0.999)
        try:
            self.load_checkpoint()
        except TapqirFileNotFoundError:
            self.iter = 0
            self.converged = False
            self._rolling = defaultdict(lambda: deque([], maxlen=100))
            self.init_parameters()
        self.elbo = self.TraceELBO(jit)
        self.nbatch_size = min(nbatch_size, self.data.Nt)
        self.fbatch_size = min(fbatch_size, self.data.F)

    # -- model.py:188-237 ---------------------------------------------------------------------------
    def run(self, num_iter: int = 0, progress_bar=None) -> None:
        use_crit = False
        if not num_iter:
            use_crit = True
            num_iter = 100000
        if progress_bar is None:
            try:
                from tqdm import tqdm as progress_bar
            except Exception:  # pragma: no cover
                progress_bar = lambda x: x
        logger.debug("Model - {}".format(self.name))
        logger.debug("Device - {}".format(self.device))
        logger.debug("Learning rate - {}".format(self.lr))
        logger.debug("AOI batch size - {}".format(self.nbatch_size))
        logger.debug("Frame batch size - {}".format(self.fbatch_size))
        writer_cm = _NullWriter()
        if SummaryWriter is not None and self.run_path is not None:
            try:
                writer_cm = SummaryWriter(log_dir=self.run_path / "logs" / self.name)
            except Exception:  # pragma: no cover
                writer_cm = _NullWriter()
        self._in_run = True
        completed = False  # run() left through its normal exit (end of the loop or convergence), not an exception
        try:
            with writer_cm as writer:
                for i in progress_bar(range(num_iter)):
                    try:
                        # checkpoint iterations need the loss on the host; the others stay asynchronous
                        if not self.iter % 200:
                            if writer is None or isinstance(writer, _NullWriter):
                                # the loss, the NaN check and the convergence parameters come to the host in ONE read-back
                                self.step_async()
                                self.save_checkpoint(None, _read_loss=True)
                            else:
                                self.iter_loss = self.step()
                                self.save_checkpoint(writer)
                            if use_crit and self.converged:
                                logger.info(f"Iteration #{self.iter} model converged.")
                                break
                        else:
                            self.step_async()
                        self.iter += 1
                    except ValueError:
                        # load last checkpoint, change rng seed (model.py:220-232)
                        self.init(lr=self.lr, nbatch_size=self.nbatch_size, fbatch_size=self.fbatch_size)
                        new_seed = random.randint(0, 100)
                        if self.collective is not None:  # every rank must draw the same global latents: rank 0 decides
                            new_seed = self.collective.broadcast_int(new_seed)
                        self.set_rng_seed(new_seed)
                        logger.warning(f"Iteration #{self.iter} restarting with a new seed: {new_seed}.")
                    except RuntimeError as err:
                        if _is_oom(err):
                            raise CudaOutOfMemoryError()
                        raise
                else:
                    logger.warning(f"Iteration #{self.iter} model has not converged.")
            completed = True
        finally:
            self._in_run = False
            if completed:
                self._join_checkpoint_writer()
                self._final_state_file()
            else:
                # unwinding from an exception (device error, KeyboardInterrupt, ...): launch nothing more and leave the
                # last good file alone -- a write here could mask the original error or store a broken state
                try:
                    self._join_checkpoint_writer()
                except Exception as err:  # pragma: no cover
                    logger.warning(f"checkpoint writer: {err!r}")
        self.iter_loss = self.last_loss()

    def _final_state_file(self):
        """End of run(): the file of the last checkpoint(s) may have been left to this point (_write_state_file).  Same
        rule as save_checkpoint -- "save only if no NaN values" (model.py:245-250): up to 199 steps have run since the
        last check, so the state is checked again (agreed between the ranks of a sharded fit: `_ckpt_file_stale` is
        the same on every rank, see _write_state_file) and a non-finite state keeps the previous file."""
        if not getattr(self, "_ckpt_file_stale", False) or self.run_path is None:
            return
        bad = not bool(torch.isfinite(self.engine.params).all())
        if self.collective is not None:
            bad = self.collective.any(bad)
        if bad:
            logger.warning(f"Iteration #{self.iter}: non-finite parameters at the end of run(); keeping the last checkpoint file")
            return
        self._write_state_file(wait=True)

    def set_rng_seed(self, seed):
        self.engine.seed = int(seed)
        self._subsample_gen = torch.Generator().manual_seed(int(seed))

    # -- model.py:239-323 ---------------------------------------------------------------------------
    def named_params(self):
        """dict name -> unconstrained leaf tensor (view into the engine's flat buffer)."""
        return self.engine.named("params")

    def save_checkpoint(self, writer=None, _read_loss=False):
        eng = self.engine
        eng.join()
        # Everything a checkpoint needs on the host in one device-to-host copy: [-ELBO, "all parameters finite", the
        # unconstrained convergence parameters].  As separate reads (loss, isfinite, one .item() per parameter) a checkpoint
        # stalled the launch queue five times -- about 1 ms per 200 iterations of 52 us each.
        views = eng.layout.views(eng.params)
        want = None if writer is not None else [n for n in self.conv_params if n != "-ELBO"]
        parts = [eng.elbo_out.reshape(1).to(torch.float64), torch.isfinite(eng.params).all().reshape(1).to(torch.float64)]
        if want is not None:
            parts += [views[n].reshape(-1).to(torch.float64) for n in want]
        host = torch.cat(parts).cpu()
        if _read_loss:
            loss = -float(host[0])
            if not math.isfinite(loss):  # as step() (model.py:212 + 220-232: run() restarts from the last checkpoint)
                raise ValueError(f"Iteration #{getattr(self, 'iter', 0)}. Non-finite loss {loss}")
            self.iter_loss = loss
        # save only if no NaN values (model.py:245-250).  In an AOI-sharded fit the ranks agree on the outcome, so that
        # all of them roll back to their checkpoints together (the local parameters live on one rank only)
        bad = float(host[1]) == 0.0
        if self.collective is not None:
            bad = self.collective.any(bad)
        if bad:
            for k, v in self.named_params().items():
                if not bool(torch.isfinite(v).all()):
                    raise ValueError("Iteration #{}. Detected NaN values in {}".format(self.iter, k))
            raise ValueError("Iteration #{}. Detected NaN values on another rank".format(self.iter))
        # (without a tensorboard writer only the convergence parameters -- globals -- are needed: transformed on the host
        # from the values read above)
        if want is None:
            cparams = eng.layout.constrained(eng.params, None)
        else:
            from torch.distributions import transform_to

            cons, cparams, at = eng.layout.constraints(), {}, 2
            for n in want:
                k = views[n].numel()
                cparams[n] = transform_to(cons[n])(host[at:at + k].reshape(views[n].shape).to(views[n].dtype))
                at += k
        for name in self.conv_params:
            if name == "-ELBO":
                self._rolling["-ELBO"].append(self.iter_loss)
            elif cparams[name].ndim == 1:
                for i in range(len(cparams[name])):
                    self._rolling[f"{name}_{i}"].append(cparams[name][i].item())
            else:
                self._rolling[name].append(cparams[name].item())
        # convergence (model.py:258-270)
        self.converged = False
        if len(self._rolling["-ELBO"]) == self._rolling["-ELBO"].maxlen:
            crit = all(
                torch.tensor(value).std() / torch.tensor(value)[-50:].std() < 1.05 for value in self._rolling.values()
            )
            if crit:
                self.converged = True
        if self.run_path is not None:
            self._write_state_file()
        if writer is not None:
            writer.add_scalar("-ELBO", self.iter_loss, self.iter)
            for name, val in cparams.items():
                if val.dim() == 0:
                    writer.add_scalar(name, val.item(), self.iter)
                elif val.dim() == 1 and len(val) <= self.Q * 2:
                    writer.add_scalars(name, {str(i): v.item() for i, v in enumerate(val)}, self.iter)
                elif val.dim() == 2 and len(val) <= self.Q * 2:
                    writer.add_scalars(
                        name, {f"{i}_{j}": k.item() for i, v in enumerate(val) for j, k in enumerate(v)}, self.iter)
        logger.debug(f"Iteration #{self.iter}: Successful.")

    # A checkpoint of a c2-sized fit is 86 MB: 9 ms to bring the three flat buffers to the host and ~32 ms for
    # torch.save, against the 13 ms that the 200 minibatch steps between two checkpoints take.  Outside run() (and on the
    # CPU) the file is written here and is complete when save_checkpoint returns.  Inside run() on a GPU the state is
    # snapshotted on the device and written by a helper process (tapqir_amd/utils/ckpt_writer.py), one file at a time: a
    # checkpoint that finds the previous file still being written leaves the file to a later one, and run() writes
    # the final state when it ends.  The bookkeeping of a checkpoint (NaN check, rolling windows, convergence) is never
    # skipped.  TAPQIR_AMD_CKPT_PROCESS=0 keeps everything in-process.
    def _manifest(self):
        eng = self.engine
        return {
            "slots": {k: (int(off), tuple(int(x) for x in shape)) for k, (off, shape) in eng.layout.slots().items()},
            "constraints": eng.layout.constraints(),
            "adam": {"step": eng.adam_step, "lr": eng.lr, "betas": tuple(eng.betas), "eps": eng.adam_eps},
            "iter": self.iter,
            "rolling": {k: list(v) for k, v in self._rolling.items()},  # model.py:279 stores the deques
            "convergence_status": self.converged,
        }

    def _write_state_file(self, wait=None):
        """Write ``<run_path>/<name>_model.tpqr`` with the current state (model.py:272-289).

        Inside run() a file may be DEFERRED (helper process busy with the previous file, or still starting).  In an
        AOI-sharded fit that decision is made collectively -- if any rank defers, all do -- so that every rank's file holds
        the same iteration: ranks resuming from files of different iterations would disagree on `iter % 200` and issue
        mismatched collectives.  Files are deferred for at most TAPQIR_AMD_CKPT_MAX_LAG seconds (default 2) in a row: a
        helper that takes longer than that is waited for, so the file on disk never falls further behind the fit.  (A
        bound by COUNT -- "the third file in a row is written synchronously" -- throttled the fit to a third of the
        writer's speed once a checkpoint interval, 200 x 54 us, had become shorter than the 40 ms a file takes.)"""
        import os
        import time

        self.run_path.mkdir(parents=True, exist_ok=True)
        target = self.run_path / f"{self.name}_model.tpqr"
        eng = self.engine
        wait = (not getattr(self, "_in_run", False)) if wait is None else wait
        w = getattr(self, "_ckpt_process", None)
        if w is not None and (w.n != eng.params.numel() or w.failed()):
            if w.failed():
                logger.warning(f"checkpoint helper process ended ({w.failure}); writing checkpoints in-process")
                os.environ["TAPQIR_AMD_CKPT_PROCESS"] = "0"
            self._ckpt_process = None
            w.close()
            w = None
        if not wait:
            defer = w is not None and (w.busy() or not w.ready())
            if self.collective is not None:
                defer = self.collective.any(defer)
            now = time.monotonic()
            since = getattr(self, "_ckpt_deferred_since", None)
            late = since is not None and now - since > float(os.environ.get("TAPQIR_AMD_CKPT_MAX_LAG", "2"))
            if self.collective is not None:
                late = self.collective.any(late)
            if defer and not late:
                if since is None:
                    self._ckpt_deferred_since = now
                self._ckpt_file_stale = True  # a later checkpoint, or the end of run(), writes a newer state
                return
            if defer:
                wait = True  # files have been left out for too long: this one is written before the fit goes on
        self._ckpt_deferred_since = None
        if w is not None and not w.failed() and (w.busy() or w.ready() or wait):
            try:
                w.submit(eng.params, eng.exp_avg, eng.exp_avg_sq, self._manifest(), target)
                if wait:
                    w.join()
                self._ckpt_file_stale = False
                return
            except RuntimeError as err:  # the helper died under this file: fall through to the in-process write
                logger.warning(f"{err}; writing checkpoints in-process")
                os.environ["TAPQIR_AMD_CKPT_PROCESS"] = "0"
                self._ckpt_process = None
                w.close()
                w = None
        if (w is None and not wait and eng.params.device.type == "cuda"
                and os.environ.get("TAPQIR_AMD_CKPT_PROCESS", "1") != "0"):
            # first file of a run(): written here, while the helper process for the following ones starts
            from tapqir_amd.utils.ckpt_writer import CheckpointWriter

            try:
                self._ckpt_process = CheckpointWriter(eng.params.numel(), eng.params.device)
            except Exception as err:  # no /dev/shm, no child process: every file is written in-process
                logger.warning(f"checkpoint helper process not available ({err!r}); writing checkpoints in-process")
                os.environ["TAPQIR_AMD_CKPT_PROCESS"] = "0"
        from tapqir_amd.utils.ckpt_writer import write_file

        payload = {
            "iter": self.iter,
            "params": self._param_store_state(),
            "optimizer": self._optim_state(),
            "rolling": {k: list(v) for k, v in self._rolling.items()},
            "convergence_status": self.converged,
        }
        write_file(payload, target)
        self._ckpt_file_stale = False

    def _join_checkpoint_writer(self, close=False):
        w = getattr(self, "_ckpt_process", None)
        if w is not None:
            if close:
                self._ckpt_process = None
                w.close()
            else:
                try:
                    w.join()
                except RuntimeError as err:  # the file in flight was lost with the helper: a later one replaces it
                    logger.warning(str(err))
                    self._ckpt_file_stale = True

    def _param_store_state(self):
        """Same payload shape as pyro.get_param_store().get_state()."""
        cons = self.engine.layout.constraints()
        # ONE download of the flat buffer; the entries are views of the host copy (a fit at the default minibatch makes
        # a step in 0.08 ms: 200 separate device-to-host copies + clones per checkpoint cost more than the 200 steps
        # between two checkpoints)
        host = self.engine.layout.views(self.engine.params.detach().cpu())
        return {"params": dict(host), "constraints": {n: cons[n] for n in host}}

    def _optim_state(self):
        """Same payload shape as pyro.optim.PyroOptim.get_state(): name -> torch Adam state_dict."""
        eng = self.engine
        m = eng.layout.views(eng.exp_avg.detach().cpu())
        v = eng.layout.views(eng.exp_avg_sq.detach().cpu())
        out = {}
        for n in m:
            out[n] = {
                "state": {0: {"step": torch.tensor(float(eng.adam_step)), "exp_avg": m[n], "exp_avg_sq": v[n]}},
                "param_groups": [{"lr": eng.lr, "betas": tuple(eng.betas), "eps": eng.adam_eps, "weight_decay": 0,
                                  "amsgrad": False, "maximize": False, "params": [0]}],
            }
        return out

    # -- model.py:325-357 ---------------------------------------------------------------------------
    def load_checkpoint(self, path: Union[str, Path] = None, param_only: bool = False, warnings: bool = False):
        self._join_checkpoint_writer()
        path = Path(path) if path else self.run_path
        if path is None:
            raise TapqirFileNotFoundError("model", f"{self.name}_model.tpqr")
        model_path = path / f"{self.name}_model.tpqr"
        try:
            checkpoint = load_tpqr(model_path, map_location="cpu")
        except FileNotFoundError:
            raise TapqirFileNotFoundError("model", model_path)
        if self.engine is None:
            self._make_engine()
        eng = self.engine
        views = eng.named("params")
        for n, t in checkpoint["params"]["params"].items():
            views[n].copy_(t.reshape(views[n].shape).to(eng.params.dtype))
        if not param_only:
            self.converged = checkpoint["convergence_status"]
            self._rolling = defaultdict(lambda: deque([], maxlen=100),
                                        {k: deque(v, maxlen=100) for k, v in checkpoint["rolling"].items()})
            self.iter = checkpoint["iter"]
            m, v = eng.named("exp_avg"), eng.named("exp_avg_sq")
            step = 0
            for n, sd in checkpoint["optimizer"].items():
                st = sd["state"][0]
                m[n].copy_(st["exp_avg"].reshape(m[n].shape).to(eng.params.dtype))
                v[n].copy_(st["exp_avg_sq"].reshape(v[n].shape).to(eng.params.dtype))
                step = int(st["step"])
            eng.reset_adam_clock(step)
            logger.info(f"Iteration #{self.iter}. Loaded a model checkpoint from {model_path}")
        if warnings and not checkpoint["convergence_status"]:
            logger.warning(f"Model at {path} has not been fully trained")

    # -- model.py:359-371 ---------------------------------------------------------------------------
    def compute_stats(self, CI: float = 0.95, save_matlab: bool = False):
        from tapqir_amd.utils.stats import save_stats

        self._join_checkpoint_writer()
        try:
            out = self.stats_path or self.path
            if out is not None:
                Path(out).mkdir(parents=True, exist_ok=True)
            save_stats(self, out, CI=CI, save_matlab=save_matlab)
        except RuntimeError as err:
            if _is_oom(err):
                raise CudaOutOfMemoryError()
            raise
        logger.debug("Computing stats: Successful.")
