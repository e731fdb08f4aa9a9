"""Drop-in surface: dataset file format, parameter layout, fit loop, checkpoints, statistics.
On the CPU the model drives the host build of the kernels' math (injected engine, tests only); the
gpu-marked test runs the same flow through libtapqir_hip.so."""

import math

import pytest
import torch

from helpers import CosmosEngine, HostCheckEngine, load_hostcheck
from tapqir_amd.exceptions import TapqirFileNotFoundError
from tapqir_amd.models import Cosmos, Model, cosmos, models
from tapqir_amd.models.layout import ParamLayout
from tapqir_amd.utils.dataset import CosmosDataset, load, save
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate


def test_registry_and_constructor_surface():
    assert models["cosmos"] is cosmos and Cosmos is cosmos and issubclass(cosmos, Model)
    m = cosmos(S=1, K=2, Q=None, device="cpu", dtype="double", use_pykeops=True)
    assert m.name == "cosmos" and m._global_params == ["gain", "proximity", "lamda", "pi"]
    assert m.conv_params == ["-ELBO", "proximity_loc", "gain_loc", "lamda_loc"]
    assert m.priors["height_std"] == 10000.0 and m.priors["gain_std"] == 50.0
    with pytest.raises(NotImplementedError):
        cosmos(S=2)


def test_data_tpqr_roundtrip(tmp_path):
    d = simulate(2, 4, 5, 1, 14, 0, TEST_PARAMS)
    save(d, tmp_path)
    raw = torch.load(tmp_path / "data.tpqr", weights_only=False)
    assert set(raw) == {"images", "xy", "is_ontarget", "mask", "labels", "offset_samples", "offset_weights", "name",
                        "time1", "ttb", "channels"}  # tapqir/utils/dataset.py:195-212
    d2 = load(tmp_path)
    assert torch.equal(d2.images, d.images) and d2.Nt == 4 and d2.F == 5 and d2.C == 1 and d2.P == 14
    assert d2.N == 2 and d2.Nc == 2 and abs(d2.offset.mean - 90.0) < 1e-4
    with pytest.raises(TapqirFileNotFoundError):
        load(tmp_path / "missing")


def test_simulated_data_follow_the_generative_law():
    d = simulate(2, 40, 50, 1, 14, 3, TEST_PARAMS)
    assert d.images.shape == (40, 50, 1, 14, 14) and torch.equal(d.images, d.images.floor())
    assert float(d.images.min()) > 90.0  # every pixel above the offset (glimpse_reader.py:407-411)
    assert int(d.is_ontarget.sum()) == 20 and (d.xy == 6.5).all()
    z = torch.as_tensor(d.labels["z"]).float()
    assert abs(float(z.mean()) - 0.15) < 0.04  # pi
    off = d.images[20:]  # off-target: background 150 + rare non-specific spots, offset 90
    assert abs(float(off.median()) - 239) < 3


def test_param_layout_is_the_reference_parameter_set():
    lay = ParamLayout(Nt=3, F=4, C=1, K=2, P=14, eps=1e-7)
    flat = torch.zeros(lay.total)
    v = lay.views(flat)
    shapes = {n: tuple(t.shape) for n, t in v.items()}
    assert shapes["m_probs"] == (2, 3, 4, 1) and shapes["b_loc"] == (3, 4, 1)
    assert shapes["background_mean_loc"] == (3, 1, 1) and shapes["pi_mean"] == (1, 2) and shapes["pi_size"] == (1, 1)
    assert set(v) == set(lay.constraints())  # cosmos.py:471-598
    assert lay.total == 18 * 12 + 2 * 3 + 9
    lay.set_constrained(flat, {"gain_loc": 5.0, "proximity_size": 100.0, "w_mean": 1.5, "pi_mean": torch.ones(1, 2) / 2})
    c = lay.constrained(flat)
    assert abs(float(c["gain_loc"]) - 5) < 1e-5 and abs(float(c["proximity_size"]) - 100) < 1e-3
    assert torch.allclose(c["w_mean"], torch.full((2, 3, 4, 1), 1.5), atol=1e-6)
    assert abs(float(flat[lay.slots()["gain_loc"][0]]) - math.log(5)) < 1e-6  # stored unconstrained


def _fit_flow(tmp_path, device, lib):
    d = simulate(2, 4, 6, 1, 14, 0, TEST_PARAMS)
    save(d, tmp_path)
    m = cosmos(K=2, device=device)
    m.load(tmp_path)
    if lib is not None:
        m._make_engine(engine_cls=HostCheckEngine)
    m.init(lr=0.005, nbatch_size=2, fbatch_size=5)
    assert m.nbatch_size == 2 and m.fbatch_size == 5 and m.iter == 0
    m.run(3, progress_bar=lambda r: r)
    assert m.iter == 3 and math.isfinite(m.iter_loss)
    ck = torch.load(tmp_path / ".tapqir" / "cosmos_model.tpqr", weights_only=False)
    assert set(ck) == {"iter", "params", "optimizer", "rolling", "convergence_status"}  # model.py:272-282
    assert set(ck["params"]) == {"params", "constraints"} and "h_loc" in ck["params"]["params"]
    assert set(ck["optimizer"]["h_loc"]["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    # a second model resumes from the checkpoint (model.py:173-180)
    m2 = cosmos(K=2, device=device)
    m2.load(tmp_path)
    if lib is not None:
        m2._make_engine(engine_cls=HostCheckEngine)
    m2.init(lr=0.005, nbatch_size=2, fbatch_size=5)
    assert m2.iter == ck["iter"]
    for n, t in ck["params"]["params"].items():
        assert torch.allclose(m2.named_params()[n].cpu().reshape(t.shape), t)
    # non-finite parameters are detected at checkpoint time (model.py:245-250)
    m2.engine.params[0] = float("nan")
    m2.iter_loss = 0.0
    with pytest.raises(ValueError):
        m2.save_checkpoint()
    return m


def test_fit_checkpoint_resume_host_build(tmp_path):
    m = _fit_flow(tmp_path, "cpu", load_hostcheck())
    assert m.m_probs.shape == (2, 4, 6, 1)


@pytest.mark.gpu
def test_fit_checkpoint_resume_and_stats_on_device(tmp_path):
    m = _fit_flow(tmp_path, "cuda", None)
    z, t = m.z_probs, m.theta_probs
    assert z.shape == (4, 6, 1, 2) and t.shape == (2, 4, 6, 1)
    assert float(z[2:].abs().max()) == 0.0  # off-target rows stay zero (cosmos.py:615-623)
    assert torch.allclose(z[:2].sum(-1), torch.ones(2, 6, 1, device=z.device), atol=1e-5)
    m.compute_stats(save_matlab=True)
    for f in ("cosmos_params.tpqr", "cosmos_summary.csv", "cosmos_params.mat"):
        assert (tmp_path / f).exists()
    p = torch.load(tmp_path / "cosmos_params.tpqr", weights_only=False)
    assert {"gain", "pi", "lamda", "proximity", "background", "height", "width", "x", "y", "m_probs", "z_probs",
            "theta_probs", "z_map", "p_specific"} <= set(p)  # cosmos.py:711-784
    assert float(p["gain"]["LL"]) < float(p["gain"]["Mean"]) < float(p["gain"]["UL"])


def test_snr_chi2_quantile_hpdi():
    """stats.py:29-86 on a noiseless image: chi2 = 0 and SNR = h sum(N^2) / sqrt(offset_var + b gain) (oracle restatement;
    the kernel is compared with it in tests/test_aux.py)."""
    from oracle.dist_util import gaussian_spots
    from oracle.stats import snr_and_chi2
    from tapqir_amd.utils.stats import hpdi, quantile

    K, F, Q, P = 2, 3, 1, 14
    h = torch.tensor([3000.0, 1500.0]).double().reshape(K, 1, 1).expand(K, F, Q)
    w = torch.full((K, F, Q), 1.4).double()
    x = torch.tensor([0.5, -3.0]).double().reshape(K, 1, 1).expand(K, F, Q)
    y = torch.tensor([-0.5, 2.0]).double().reshape(K, 1, 1).expand(K, F, Q)
    tl = torch.full((F, Q, 2), 6.5).double()
    b = torch.full((F, Q), 150.0).double()
    g = gaussian_spots(h, w, x, y, tl, P)
    data = b[..., None, None] + g.sum(0) + 90.0
    snr, chi2 = snr_and_chi2(data, h, w, x, y, tl, b, torch.tensor(7.0).double(), 90.0, 4.0, P)
    assert chi2.abs().max() < 1e-20
    wk = g / h[..., None, None]
    expect = (g.sum(0)[None] * wk).sum((-1, -2)) / (4.0 + 150.0 * 7.0) ** 0.5
    assert torch.allclose(snr, expect, rtol=1e-12)
    s = torch.tensor([3.0, 1.0, 2.0, 10.0, 4.0])
    assert float(quantile(s, 0.5)) == 3.0 and float(quantile(s, 0.25)) == 2.0
    lo, hi = hpdi(torch.tensor([0.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]), 0.8)
    assert (float(lo), float(hi)) == (1.0, 1.0)


@pytest.mark.gpu
def test_parallel_attach_single_rank_group(tmp_path):
    """tapqir_amd.parallel.attach on a one-rank RCCL group: the sharded code path (staged step, asynchronous all-reduce,
    deferred global tail) gives the same parameters as the plain single-process fit."""
    import os

    import torch.distributed as dist

    from tapqir_amd.models import models
    from tapqir_amd.parallel import attach
    from tapqir_amd.utils.dataset import save
    from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

    save(simulate(2, 4, 8, 1, 14, seed=1, params=TEST_PARAMS), tmp_path)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29631")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        outs = []
        for sharded in (False, True):
            m = models["cosmos"](S=1, K=2, device="cuda", dtype="float")
            m.load(tmp_path)
            if sharded:
                attach(m)
            m.init(lr=0.005, nbatch_size=4, fbatch_size=8)
            m.run(5, progress_bar=lambda x: x)
            assert torch.isfinite(m.engine.params).all()
            outs.append((m.engine.params.clone(), m.iter_loss))
            (tmp_path / ".tapqir" / "cosmos_model.tpqr").unlink()
        assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-6)
        assert abs(outs[0][1] - outs[1][1]) <= 1e-6 * abs(outs[0][1])
        # minibatch steps (lazy Adam + deferred tail behind the asynchronous all-reduce) against the plain fit
        outs = []
        for sharded in (False, True):
            m = models["cosmos"](S=1, K=2, device="cuda", dtype="float")
            m.load(tmp_path)
            if sharded:
                attach(m)
            m.init(lr=0.005, nbatch_size=2, fbatch_size=3)
            m.run(12, progress_bar=lambda x: x)
            assert m.engine.lazy_adam
            outs.append((m.engine.params.clone(), m.engine.exp_avg_sq.clone(), m.iter_loss))
            (tmp_path / ".tapqir" / "cosmos_model.tpqr").unlink()
        assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-6)
        assert torch.allclose(outs[0][1], outs[1][1], rtol=1e-4, atol=1e-12)
        assert abs(outs[0][2] - outs[1][2]) <= 1e-6 * abs(outs[0][2])
    finally:
        dist.destroy_process_group()


def test_data_median_is_the_full_data_median():
    """Initial background = median of ALL pixels (dataset.py:134-138), not of the head of the buffer: a dataset whose
    first AOIs are bright must not shift it; the counting median of integer data equals torch.median exactly."""
    from tapqir_amd.models.cosmos import data_median

    g = torch.Generator().manual_seed(0)
    img = torch.randint(100, 400, (6, 5, 2, 4, 4), generator=g).float()
    img[:2] += 5000.0  # bright head
    want = torch.stack([torch.median(img[:, :, c]) for c in range(2)]).double()
    assert torch.equal(data_median(img), want)
    assert torch.equal(data_median(img, chunk=37), want)
    # non-integer data: plain torch.median
    imf = img + 0.25
    assert torch.equal(data_median(imf), torch.stack([torch.median(imf[:, :, c]) for c in range(2)]).double())
    # even number of pixels: the LOWER middle value, as torch.median
    ev = torch.tensor([1.0, 2.0, 3.0, 4.0]).reshape(1, 1, 1, 2, 2)
    assert float(data_median(ev)[0]) == 2.0


def test_restricted_loader_refuses_foreign_classes(tmp_path):
    """.tpqr files are read with torch.load(weights_only=True) + an allow-list; a pickle that names any other class is
    refused instead of executed."""
    import pickle

    from tapqir_amd.utils.safe_load import load_tpqr

    class Evil:
        def __reduce__(self):
            return (print, ("executed code from the file",))

    torch.save({"images": torch.zeros(1), "x": Evil()}, tmp_path / "bad.tpqr")
    with pytest.raises(pickle.UnpicklingError):
        load_tpqr(tmp_path / "bad.tpqr")
    import collections
    torch.save({"rolling": {"a": collections.deque([1.0, 2.0], maxlen=100)}}, tmp_path / "ref_style.tpqr")
    with pytest.raises(pickle.UnpicklingError, match="TAPQIR_AMD_TRUST_FILES"):
        load_tpqr(tmp_path / "ref_style.tpqr")


def test_stats_with_on_and_off_target_labels(tmp_path):
    """read_glimpse stacks off-target labels behind the on-target ones (Nt rows); the classification statistics use the
    first N rows (stats.py:196, 217)."""
    import numpy as np

    d = simulate(2, 4, 6, 1, 14, 0, TEST_PARAMS)
    off = np.zeros((2, 6, 1), dtype=d.labels.dtype)
    d.labels = np.concatenate([d.labels, off], 0)
    assert d.labels.shape[0] == d.Nt
    save(d, tmp_path)
    m = cosmos(K=2, device="cpu")
    m.load(tmp_path)
    m._make_engine(engine_cls=HostCheckEngine)
    m.init(lr=0.005, nbatch_size=4, fbatch_size=6)
    m.run(2, progress_bar=lambda r: r)
    m.compute_stats()
    assert "MCC" in m.summary.index and (tmp_path / "cosmos_summary.csv").is_file()


def test_simulate_kinetic_mode():
    """simulate.py:66-86: with kon / koff the labels follow a two-state Markov chain over the frames."""
    p = dict(TEST_PARAMS)
    del p["pi"]
    p.update(kon=0.05, koff=0.2)
    d = simulate(2, 80, 400, 1, 14, 0, p)
    z = torch.from_numpy(d.labels["z"][..., 0]).double()  # (N_on, F)
    assert z.shape == (40, 400)
    assert abs(float(z.mean()) - 0.05 / 0.25) < 0.02  # stationary occupancy kon / (kon + koff)
    on_to_off = ((z[:, :-1] == 1) & (z[:, 1:] == 0)).double().sum() / (z[:, :-1] == 1).double().sum()
    off_to_on = ((z[:, :-1] == 0) & (z[:, 1:] == 1)).double().sum() / (z[:, :-1] == 0).double().sum()
    assert abs(float(on_to_off) - 0.2) < 0.02 and abs(float(off_to_on) - 0.05) < 0.006
    # frames with a target-specific spot are brighter at the target
    img = d.images[:40, :, 0, 6:8, 6:8].mean((-1, -2))
    assert float(img[z == 1].mean()) > float(img[z == 0].mean()) + 50
    # the explicit init / trans spelling (simulate.py:57-64) gives the same chain law
    p2 = dict(TEST_PARAMS)
    del p2["pi"]
    p2.update(init=torch.tensor([[0.8, 0.2]]), trans=torch.tensor([[[0.95, 0.05], [0.2, 0.8]]]))
    d2 = simulate(2, 80, 400, 1, 14, 0, p2)
    assert torch.equal(torch.from_numpy(d2.labels["z"]), torch.from_numpy(d.labels["z"]))
    with pytest.raises(ValueError):
        simulate(2, 4, 4, 1, 14, 0, {k: v for k, v in TEST_PARAMS.items() if k != "pi"})


def test_checkpoint_writer_process_writes_the_in_process_payload(tmp_path):
    """The helper process of run() (tapqir_amd/utils/ckpt_writer.py) writes the file the in-process path writes."""
    from tapqir_amd.utils.ckpt_writer import CheckpointWriter, build_payload
    from tapqir_amd.utils.safe_load import load_tpqr

    lay = ParamLayout(Nt=3, F=4, C=1, K=2, P=14, eps=1e-7)
    n = lay.total
    g = torch.Generator().manual_seed(3)
    p, m1, m2 = (torch.randn(n, generator=g) for _ in range(3))
    manifest = {"slots": {k: (int(o), tuple(s)) for k, (o, s) in lay.slots().items()}, "constraints": lay.constraints(),
                "adam": {"step": 17, "lr": 0.005, "betas": (0.9, 0.999), "eps": 1e-8}, "iter": 200,
                "rolling": {"-ELBO": [3.0, 2.0]}, "convergence_status": False}
    w = CheckpointWriter(n, "cpu")
    try:
        target = tmp_path / "cosmos_model.tpqr"
        w.submit(p, m1, m2, manifest, target)
        p_later = p.clone()
        w.join()
        assert w.files_written == 1 and not w.busy()
        ck = load_tpqr(target, map_location="cpu")
        want = build_payload(torch.cat([p_later, m1, m2]), dict(manifest, n=n))
        assert set(ck) == set(want) and ck["iter"] == 200 and ck["rolling"] == {"-ELBO": [3.0, 2.0]}
        for name, t in want["params"]["params"].items():
            assert torch.equal(ck["params"]["params"][name], t) and t.shape == lay.slots()[name][1]
            st, st_w = ck["optimizer"][name]["state"][0], want["optimizer"][name]["state"][0]
            assert torch.equal(st["exp_avg"], st_w["exp_avg"]) and torch.equal(st["exp_avg_sq"], st_w["exp_avg_sq"])
            assert float(st["step"]) == 17.0
        assert str(ck["params"]["constraints"]["w_mean"]) == str(lay.constraints()["w_mean"])
        # a second file through the same process; an unwritable target is reported at the next join
        w.submit(p + 1, m1, m2, dict(manifest, iter=400), target)
        w.join()
        assert load_tpqr(target, map_location="cpu")["iter"] == 400
        w.submit(p, m1, m2, manifest, tmp_path / "missing_dir" / "x.tpqr")
        with pytest.raises(RuntimeError):
            w.join()
    finally:
        w.close()
    import os
    assert not os.path.exists(w.path)


@pytest.mark.gpu
def test_run_leaves_the_final_state_on_disk(tmp_path):
    """Inside run() the checkpoint files are written by the helper process, one at a time; whatever was skipped while it
    was busy, the file on disk after run() holds the state and iteration count the model ended with."""
    d = simulate(2, 4, 40, 1, 14, 0, TEST_PARAMS)
    save(d, tmp_path)
    m = cosmos(K=2, device="cuda")
    m.load(tmp_path)
    m.init(lr=0.005, nbatch_size=2, fbatch_size=16)
    m.run(450, progress_bar=lambda r: r)
    ck = torch.load(tmp_path / ".tapqir" / "cosmos_model.tpqr", weights_only=False)
    assert ck["iter"] in (400, 450)
    if ck["iter"] == 450:  # the files of iterations 200 / 400 found the writer busy: run() wrote the final state
        for n, t in ck["params"]["params"].items():
            assert torch.equal(m.named_params()[n].cpu().reshape(t.shape), t)
    assert len(ck["rolling"]["-ELBO"]) == 3
    m2 = cosmos(K=2, device="cuda")
    m2.load(tmp_path)
    m2.init(lr=0.005, nbatch_size=2, fbatch_size=16)
    assert m2.iter == ck["iter"] and m2.engine.adam_step == m.engine.adam_step - (450 - ck["iter"])
    import os
    for model in (m, m2):
        w = getattr(model, "_ckpt_process", None)
        model._join_checkpoint_writer(close=True)
        assert w is None or not os.path.exists(w.path)


class _FakeWriter:
    """Stand-in for CheckpointWriter in the host-side tests of Model._write_state_file (no GPU, no child process)."""

    def __init__(self, n, busy=False, ready=True, dead=False):
        self.n, self._busy, self._rdy, self.failure = n, busy, ready, ("exit code 1" if dead else None)
        self.submitted, self.closed = [], False

    def busy(self):
        return self._busy

    def ready(self):
        return self._rdy

    def failed(self):
        return self.failure is not None

    def submit(self, p, m, v, manifest, target):
        self.submitted.append(manifest["iter"])

    def join(self):
        pass

    def close(self):
        self.closed = True


def _host_model(tmp_path, iters=1):
    d = simulate(2, 4, 6, 1, 14, 0, TEST_PARAMS)
    save(d, tmp_path)
    m = cosmos(K=2, device="cpu")
    m.load(tmp_path)
    m._make_engine(engine_cls=HostCheckEngine)
    m.init(lr=0.005, nbatch_size=2, fbatch_size=5)
    m.run(iters, progress_bar=lambda r: r)
    return m


def test_checkpoint_files_deferred_for_a_bounded_time_and_dead_helper_is_dropped(tmp_path, monkeypatch):
    """a helper process that died must not turn every later checkpoint into a deferred one; a busy helper
    defers files for at most TAPQIR_AMD_CKPT_MAX_LAG seconds in a row (a bound by count would tie the fit to the writer once
    200 iterations take less time than a file)."""
    import time

    monkeypatch.setenv("TAPQIR_AMD_CKPT_MAX_LAG", "0.3")
    m = _host_model(tmp_path)
    target = tmp_path / ".tapqir" / "cosmos_model.tpqr"
    n = m.engine.params.numel()
    m._in_run = True
    # busy helper: deferrals while the lag is short (any number of them), then a file is submitted and waited for
    w = m._ckpt_process = _FakeWriter(n, busy=True)
    for it in (200, 400, 600, 800):
        m.iter = it
        m._write_state_file()
        assert m._ckpt_file_stale is True and len(w.submitted) == 0
    time.sleep(0.35)
    m.iter = 1000
    m._write_state_file()
    assert m._ckpt_file_stale is False and w.submitted == [1000]
    # ... and the clock starts again with the next deferral
    m.iter = 1200
    m._write_state_file()
    assert m._ckpt_file_stale is True and w.submitted == [1000]
    # a helper that is gone: dropped, the file is written in-process at once
    m._ckpt_process = dead = _FakeWriter(n, dead=True)
    m.iter = 1400
    target.unlink()
    m._write_state_file()
    assert dead.closed and m._ckpt_process is None and not m._ckpt_file_stale
    assert torch.load(target, weights_only=False)["iter"] == 1400
    m._in_run = False


def test_dead_helper_process_is_detected(tmp_path):
    """The real CheckpointWriter with a child that exits at once (what an import failure or an OOM kill looks like)."""
    import subprocess
    import sys
    import time

    from tapqir_amd.utils.ckpt_writer import CheckpointWriter

    w = CheckpointWriter(16, "cpu")
    try:
        assert not w.failed()
        w.child.kill()
        w.child.wait()
        t0 = time.time()
        while not w.failed() and time.time() - t0 < 5:
            time.sleep(0.01)
        assert w.failed() and w.failure
        assert not w.ready()  # EOF on the pipe is not "ready"
    finally:
        w.close()


def test_run_keeps_the_last_file_when_the_final_state_is_not_finite(tmp_path):
    """model.py:245-250 "save only if no NaN values" also holds for the file run() writes when it ends."""
    m = _host_model(tmp_path, iters=3)
    target = tmp_path / ".tapqir" / "cosmos_model.tpqr"
    before = torch.load(target, weights_only=False)
    m._ckpt_file_stale = True
    m.engine.params[0] = float("nan")
    m._final_state_file()
    after = torch.load(target, weights_only=False)
    assert after["iter"] == before["iter"]
    assert all(bool(torch.isfinite(t).all()) for t in after["params"]["params"].values())
    # and a finite state IS written
    m.engine.params[0] = 0.0
    m.iter = 77
    m._final_state_file()
    assert torch.load(target, weights_only=False)["iter"] == 77 and not m._ckpt_file_stale


def test_run_unwinding_from_an_exception_writes_nothing(tmp_path):
    m = _host_model(tmp_path, iters=1)
    target = tmp_path / ".tapqir" / "cosmos_model.tpqr"
    before = torch.load(target, weights_only=False)["iter"]
    m._ckpt_file_stale = True

    def boom():
        raise KeyboardInterrupt

    m.step_async = boom
    m.iter = 1  # not a checkpoint iteration: run() goes through step_async
    with pytest.raises(KeyboardInterrupt):
        m.run(5, progress_bar=lambda r: r)
    assert torch.load(target, weights_only=False)["iter"] == before
