"""Input side: ``tapqir glimpse`` preprocessing.  CPU: the oracle against closed-form cases, the host
parser against the oracle's independent parser, histogram post-processing, ABI argument checks.  GPU: read_glimpse
(tq_glimpse_extract) against the oracle, bit for bit."""

import ctypes as C

import numpy as np
import pytest
import torch

from glimpse_fixture import analytic_frame, write_experiment
from oracle import glimpse as og
from tapqir_amd import _lib
from tapqir_amd.exceptions import HipExtensionError
from tapqir_amd.imscroll import GlimpseDataset, bin_hist, read_glimpse
from tapqir_amd.imscroll import glimpse_reader as gr


def channel_kwargs(cfg, c=0):
    kw = {k: v for k, v in cfg.items() if k not in ("P", "num-channels", "dataset", "channels", "offset-P", "bin-size")}
    return {**kw, **cfg["channels"][c]}


# ---- oracle against closed forms ----------------------------------------------------------------------------------
def test_oracle_decode_and_crop_closed_form(tmp_path):
    cfg, truth = write_experiment(tmp_path, kind="analytic", F=9, labels=False)
    out = og.read_glimpse(**cfg)
    H, W = truth[0].shape[1:]
    header = og.read_header(cfg["channels"][0]["glimpse-folder"])
    for f in (1, 5, 9):  # decode: big-endian int16 + 2^15 returns the written counts, across both files and the gaps
        assert np.array_equal(og.read_frame(cfg["channels"][0]["glimpse-folder"], header, f), truth[0][f - 1])
    P = cfg["P"]
    xy, images = out["xy"].numpy(), out["images"].numpy()
    assert images.shape == (8, 9, 1, P, P) and images.dtype == np.int64
    assert ((xy > 0.5 * P - 1) & (xy < 0.5 * P)).all()
    # the value of a pixel encodes its frame position: recover the crop corner of every AOI-frame from pixel (0, 0)
    # and check the whole window and the target position against it
    frames, dxy = og.read_driftlist(cfg["channels"][0]["driftlist"])
    cum = og.cumulative_drift(frames, dxy, 4)
    for d, rows in (("ontarget", slice(0, 5)), ("offtarget", slice(5, 8))):
        _, x, y, _ = og.read_aoiinfo(cfg["channels"][0][f"{d}-aoiinfo"])
        raw = np.stack([x, y], -1)[:, None] + cum[None]
        corner = np.rint(raw - xy[rows, :, 0])
        assert np.allclose(raw - corner, xy[rows, :, 0], atol=1e-12)
        for n in range(raw.shape[0]):
            for f in range(9):
                cx, cy = int(corner[n, f, 0]), int(corner[n, f, 1])
                assert np.array_equal(images[rows][n, f, 0], analytic_frame(f, H, W)[cy:cy + P, cx:cx + P])


def test_oracle_cumulative_drift_known_answer():
    frames = np.arange(3, 10)  # aoiinfo frame 6 = position 3
    d = np.stack([np.arange(1.0, 8.0), 10 * np.arange(1.0, 8.0)], -1)
    cum = og.cumulative_drift(frames, d, 6)
    # forwards: running sum from frame 7; backwards: minus the drift of frames f+1..6; frame 6 keeps its entry
    assert cum[:, 0].tolist() == [-(2 + 3 + 4), -(3 + 4), -4, 4, 5, 5 + 6, 5 + 6 + 7]
    assert np.array_equal(cum[:, 1], 10 * cum[:, 0])
    with pytest.raises(ValueError):
        og.cumulative_drift(frames, d, 42)


def test_bin_hist_known_answers_and_product_matches_oracle():
    s = torch.arange(100, 111, dtype=torch.int)  # 11 samples: first + 3 runs of 3 + tail of 1
    w = torch.arange(1, 12, dtype=torch.float64) / 66
    bs, bw = bin_hist(s, w, 3)
    assert bs.tolist() == [100, 102, 105, 108, 110]
    assert torch.allclose(bw.double(), torch.tensor([1, 2 + 3 + 4, 5 + 6 + 7, 8 + 9 + 10, 11], dtype=torch.float64) / 66.0, atol=1e-7)
    assert bs.dtype == torch.int32 and bw.dtype == torch.float32
    bs1, bw1 = bin_hist(s, w, 1)
    assert bs1.tolist() == s.tolist() and torch.equal(bw1, w.float())
    gen = torch.Generator().manual_seed(0)
    for n in (1, 2, 5, 16, 37, 64):
        for size in (1, 2, 3, 4, 5, 7, 21):
            samples = torch.cumsum(torch.randint(1, 4, (n,), generator=gen), 0).int()
            weights = torch.rand(n, generator=gen, dtype=torch.float64)
            weights /= weights.sum()
            got, want = bin_hist(samples, weights, size), og.bin_hist(samples, weights, size)
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), (n, size)
            assert got[0].dtype == want[0].dtype and got[1].dtype == want[1].dtype


def test_offset_postprocessing_matches_oracle():
    rng = np.random.default_rng(1)
    for min_data, bin_size in ((50, 1), (88, 3), (97, 5), (200, 1)):
        hist = np.zeros(65536, dtype=np.int64)
        hist[80:120] = rng.integers(1, 5000, 40)
        hist[[300, 4000, 65535]] = [3, 1, 2]  # the sparse top: folded into the last kept sample
        counts = {int(v): int(hist[v]) for v in np.flatnonzero(hist)}
        got = gr._finish_offsets(hist, min_data, bin_size)
        want = og.finish_offsets(counts, min_data, bin_size)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
        assert abs(float(got[1].sum()) - 1) < 1e-6
        assert (got[0][0].item() == min_data - 1) == (min_data <= 80)
    with pytest.raises(ValueError):
        gr._finish_offsets(np.zeros(65536, dtype=np.int64), 10, 1)


# ---- host parser (pandas) against the oracle's numpy parser -------------------------------------------------------
@pytest.mark.parametrize("frame_range", [None, (3, 9)])
def test_glimpse_dataset_metadata_matches_oracle(tmp_path, frame_range):
    cfg, truth = write_experiment(tmp_path, C=2, frame_range=frame_range)
    for c in range(2):
        ch = cfg["channels"][c]
        g = GlimpseDataset(**channel_kwargs(cfg, c), c=c)
        frames, dxy = og.read_driftlist(ch["driftlist"])
        cum = og.cumulative_drift(frames, dxy, 4)
        keep = np.ones(len(frames), bool) if frame_range is None else (frames >= 3) & (frames <= 9)
        assert np.array_equal(g.cumdrift.index.values, frames[keep])
        assert np.array_equal(g.cumdrift[["dx", "dy"]].values, cum[keep])  # bit-exact: same summation order
        assert np.array_equal(g.cumdrift["ttb"].values, (1000.0 * np.arange(12) + 17 * c)[keep])
        assert (g.height, g.width, g.N, g.Nc, g.F, len(g)) == (64, 72, 5, 3, keep.sum(), keep.sum())
        assert g.dtypes == ["ontarget", "offtarget"] and "N=5, Nc=3" in repr(g)
        for d in g.dtypes:
            aoi, x, y, _ = og.read_aoiinfo(ch[f"{d}-aoiinfo"])
            assert np.array_equal(g.aoiinfo[d].index.values, aoi)
            assert np.array_equal(g.aoiinfo[d]["x"].values, x) and np.array_equal(g.aoiinfo[d]["y"].values, y)
        lab = og.spotpicker_labels(ch["ontarget-labels"], g.aoiinfo["ontarget"].index.values, frames[keep])
        assert np.array_equal(g.labels["ontarget"], lab) and g.labels["offtarget"] is None
        assert lab["z"][0].tolist() == [(4 <= f <= 7) for f in frames[keep]]  # AOI 1: present in frames 4..7
        assert lab["z"][3].tolist() == [f == 2 for f in frames[keep]]
        # host-side frame access (plots, inspection) decodes like the oracle
        assert np.array_equal(g[int(frames[keep][0])], truth[c][int(frames[keep][0]) - 1])
        assert np.array_equal(g[3:6], truth[c][2:5])
    g0 = GlimpseDataset(**channel_kwargs(cfg, 0))
    raw = np.empty(3 * 2 * 64 * 72, dtype=np.uint8)
    g0.read_raw([6, 7, 8], raw)  # file 1, with junk between frames
    assert np.array_equal(raw.view(">i2").astype(np.int64).reshape(3, 64, 72) + 2 ** 15, truth[0][5:8])


# ---- no CPU product path ---------------------------------------------------------------------------------------------
def test_read_glimpse_refuses_to_run_without_a_gpu(tmp_path):
    cfg, _ = write_experiment(tmp_path, F=4, labels=False)
    with pytest.raises(HipExtensionError):
        read_glimpse(tmp_path, None, device="cpu", **cfg)


def test_glimpse_abi_argument_checks():
    lib = _lib.load()
    assert lib.tq_glimpse_extract(None, None) == 1
    a = _lib.GlimpseArgs(H=64, W=64, N=1, F=4, C=1, P=14, c=0, f0=0, nf=4)
    assert lib.tq_glimpse_extract(C.byref(a), None) == 1 and b"NULL" in lib.tq_last_error()
    a = _lib.GlimpseArgs(frames=8, raw_xy=8, images=8, target_xy=8, status=8, H=64, W=64, N=1, F=4, C=1, P=14, c=0, f0=2, nf=4)
    assert lib.tq_glimpse_extract(C.byref(a), None) == 1 and b"geometry" in lib.tq_last_error()  # f0 + nf > F
    a.f0, a.P = 0, 65
    assert lib.tq_glimpse_extract(C.byref(a), None) == 1  # window larger than the frame


# ---- GPU parity ---------------------------------------------------------------------------------------------------------
def assert_same_dataset(ds, want):
    assert ds.images.dtype == torch.int64 and torch.equal(ds.images, want["images"])
    assert ds.xy.dtype == torch.float64 and torch.equal(ds.xy, want["xy"])
    assert torch.equal(ds.is_ontarget, want["is_ontarget"])
    assert torch.equal(ds.offset.samples.cpu(), want["offset_samples"]) and ds.offset.samples.dtype == torch.int32
    assert torch.equal(ds.offset.weights.cpu(), want["offset_weights"])
    assert torch.equal(ds.time1, want["time1"]) and torch.equal(ds.ttb, want["ttb"])
    if want["labels"] is None:
        assert ds.labels is None
    else:
        assert np.array_equal(ds.labels, want["labels"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    dict(), dict(C=2, P=9, bin_size=3), dict(frame_range=(2, 10), n_off=0, labels=False), dict(P=21, H=80, W=70, bin_size=5), dict(P=20, H=80, W=70, labels=False), dict(P=6, n_on=9),
    dict(F=1, n_on=1, n_off=0, labels=False, aoiinfo_frame=1),
])
def test_read_glimpse_matches_oracle(tmp_path, monkeypatch, case):
    case = dict(case)
    bin_size = case.pop("bin_size", 1)
    cfg, _ = write_experiment(tmp_path / "raw", seed=3, **case)
    cfg["bin-size"] = bin_size
    monkeypatch.setattr(gr, "CHUNK_BYTES", 5 * 2 * 64 * 72)  # several chunks, both pinned buffers reused
    ds = read_glimpse(tmp_path, None, **cfg)
    want = og.read_glimpse(**cfg)
    assert_same_dataset(ds, want)
    # data.tpqr holds the same thing, under the reference's keys (dataset.py:195-212)
    saved = torch.load(tmp_path / "data.tpqr", weights_only=False)
    assert set(saved) == {"images", "xy", "is_ontarget", "mask", "labels", "offset_samples", "offset_weights", "name",
                          "time1", "ttb", "channels"}
    assert torch.equal(saved["images"], want["images"]) and saved["name"] == "synthetic"
    assert saved["channels"] == tuple(f"dye{c}" for c in range(cfg["num-channels"]))


@pytest.mark.gpu
def test_extract_kernel_ties_extremes_and_windows_on_the_border():
    """Straight through the C ABI: positions whose corner is exactly half-way between pixels (round half to even),
    windows flush with every frame border, pixel extremes 0 and 65535, and windows that leave the frame."""
    lib = _lib.load()
    rng = np.random.default_rng(5)
    H, W, P, nf, F, f0 = 40, 48, 14, 3, 5, 1
    frames = rng.integers(0, 65536, size=(nf, H, W)).astype(np.int64)
    frames[0, 0, 0], frames[1, H - 1, W - 1] = 0, 65535
    half = 0.5 * (P - 1)
    xs = [half + 0.5, half + 1.5, half + 2.5, half, W - P + half, half + 10.2, half - 0.5000001, W - P + half + 1.5]
    ys = [half + 3.5, half + 4.5, half, H - P + half, half + 7.49999, half + 0.5, half + 2.0, half]
    N = len(xs)
    raw = np.zeros((N, nf, 2))
    raw[:, :, 0], raw[:, :, 1] = np.array(xs)[:, None], np.array(ys)[:, None]
    raw[5, 1] += [1e-9, -1e-9]
    dev = torch.device("cuda")
    d_frames = torch.from_numpy((frames - 2 ** 15).astype(">i2").view(np.uint8).reshape(-1).copy()).to(dev)
    d_raw = torch.from_numpy(raw).to(dev)
    images = torch.full((N, F, 1, P, P), -7, dtype=torch.int32, device=dev)
    txy = torch.full((N, F, 1, 2), -7.0, dtype=torch.float64, device=dev)
    status = torch.tensor([0, 2 ** 31 - 1], dtype=torch.int32, device=dev)
    a = _lib.GlimpseArgs(frames=d_frames.data_ptr(), raw_xy=d_raw.data_ptr(), images=images.data_ptr(), target_xy=txy.data_ptr(),
                         offset_hist=None, status=status.data_ptr(), H=H, W=W, N=N, F=F, C=1, P=P, c=0, f0=f0, nf=nf)
    _lib.check(lib.tq_glimpse_extract(a, torch.cuda.current_stream().cuda_stream), "tq_glimpse_extract")
    torch.cuda.synchronize()
    got, gxy = images.cpu().numpy(), txy.cpu().numpy()
    inside = [n for n in range(N) if n not in (6, 7)]  # 6 starts at -1 (half - 0.5000001 rounds to -1), 7 ends past W
    want = np.zeros((N, nf, P, P), dtype=np.int64)
    wxy = np.zeros((N, nf, 2))
    for f in range(nf):
        sub_i, sub_xy = np.zeros((len(inside), P, P), dtype=np.int64), np.zeros((len(inside), 2))
        og.extract_frame(frames[f], raw[inside, f], P, sub_i, sub_xy)
        want[inside, f], wxy[inside, f] = sub_i, sub_xy
    assert np.array_equal(got[inside][:, f0:f0 + nf, 0], want[inside])
    assert np.array_equal(gxy[inside][:, f0:f0 + nf, 0], wxy[inside])
    assert (got[:, [0, 4]] == -7).all() and (got[[6, 7]] == -7).all()  # frames outside the chunk / skipped windows untouched
    assert status.tolist() == [2 * nf, int(want[inside].min())]
    for bad in (6, 7):  # the reference cannot extract these: numpy slicing yields a short or empty window
        with pytest.raises(ValueError):
            og.extract_frame(frames[0], raw[[bad], 0], P, np.zeros((1, P, P), dtype=np.int64), np.zeros((1, 2)))
    # round half to even picked the even corner for both tie cases
    assert (gxy[0, f0, 0, 0], gxy[1, f0, 0, 0]) == (half + 0.5, half - 0.5)  # corners 0 and 2


@pytest.mark.gpu
def test_read_glimpse_raises_like_the_reference_when_a_window_leaves_the_frame(tmp_path):
    cfg, _ = write_experiment(tmp_path / "raw", labels=False, drift_scale=4.0, F=40, seed=11)
    with pytest.raises(ValueError):
        og.read_glimpse(**cfg)
    with pytest.raises(ValueError, match="leave"):
        read_glimpse(tmp_path, None, **cfg)


@pytest.mark.gpu
def test_read_glimpse_full_size_closed_form(tmp_path):
    """512 x 512 frames, 300 frames, 600 AOIs (35 M extracted pixels): every pixel against the position-encoding
    frame formula, evaluated with the corner recovered from the returned target positions."""
    H = W = 512
    F, n_on, n_off, P = 300, 400, 200, 14
    cfg, _ = write_experiment(tmp_path / "raw", H=H, W=W, F=F, n_on=n_on, n_off=n_off, P=P, kind="analytic", labels=False,
                              drift_scale=0.05, aoiinfo_frame=150)
    ds = read_glimpse(tmp_path, None, **cfg)
    frames, dxy = og.read_driftlist(cfg["channels"][0]["driftlist"])
    cum = og.cumulative_drift(frames, dxy, 150)
    xs, ys = [], []
    for d in ("ontarget", "offtarget"):
        _, x, y, _ = og.read_aoiinfo(cfg["channels"][0][f"{d}-aoiinfo"])
        xs.append(x), ys.append(y)
    raw = np.stack([np.concatenate(xs), np.concatenate(ys)], -1)[:, None] + cum[None]
    corner = torch.from_numpy(np.rint(raw - 0.5 * (P - 1))).long()
    assert torch.equal(ds.xy[:, :, 0], torch.from_numpy(raw) - corner)
    i = torch.arange(P)
    rows = corner[..., 1, None, None] + i[:, None]
    cols = corner[..., 0, None, None] + i[None, :]
    want = (rows * 977 + cols * 31 + torch.arange(F)[None, :, None, None] * 7919) % 65536
    assert torch.equal(ds.images[:, :, 0], want)
    # offset region of the analytic frames: exact counts
    counts = {}
    for f in range(F):
        og.count_offsets(analytic_frame(f, H, W), cfg["offset-x"], cfg["offset-y"], cfg["offset-P"], counts)
    s, w = og.finish_offsets(counts, int(want.min()), 1)
    assert torch.equal(ds.offset.samples.cpu(), s) and torch.equal(ds.offset.weights.cpu(), w)


@pytest.mark.gpu
def test_frames_to_fit(tmp_path):
    """Whole input side feeding the hot path: simulated AOI images are painted into drifting full frames, extracted with
    read_glimpse (GPU), and the fit of the extracted data.tpqr recovers the simulated binding labels."""
    import math

    from glimpse_fixture import write_frames_experiment
    from tapqir_amd.models import models
    from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

    N, F, P = 16, 150, 14
    sim = simulate(2, N, F, 1, P, seed=4, params=TEST_PARAMS)
    rng = np.random.default_rng(0)
    H = W = 160
    step = rng.integers(-1, 2, size=(F, 2))  # integer stage drift per frame
    ref = 40  # aoiinfo frame (1-based): its own drift entry stays raw, so it is zero here
    step[ref - 1] = 0
    cum = og.cumulative_drift(np.arange(1, F + 1), step.astype(float), ref)
    assert np.abs(cum).max() < 18
    # AOI corners on a grid with room for the drift; the target sits at the window centre (P - 1) / 2 like the simulation
    gx, gy = np.meshgrid(np.arange(4), np.arange(4))
    corner = np.stack([30 + 32 * gx.ravel(), 30 + 32 * gy.ravel()], -1)[:N]
    centre = corner + 0.5 * (P - 1)
    frames = rng.integers(88, 93, size=(F, H, W)).astype(np.int64)  # camera offset ~ 90 everywhere else
    imgs = sim.images[:, :, 0].numpy().astype(np.int64)
    for f in range(F):
        for n in range(N):
            sx, sy = int(corner[n, 0] + cum[f, 0]), int(corner[n, 1] + cum[f, 1])
            frames[f, sy:sy + P, sx:sx + P] = imgs[n, f]
    cfg = write_frames_experiment(str(tmp_path / "raw"), frames, centre[: N // 2], centre[N // 2:], step.astype(float), ref, P,
                                  offset_x=0, offset_y=0, offset_P=10)
    ds = read_glimpse(tmp_path, None, **cfg)
    assert torch.equal(ds.images[:, :, 0], torch.from_numpy(imgs))  # the painted tiles come back bit for bit
    assert torch.equal(ds.xy, torch.full((N, F, 1, 2), 0.5 * (P - 1), dtype=torch.float64))
    assert ds.is_ontarget.tolist() == [True] * (N // 2) + [False] * (N // 2)
    assert 86 <= int(ds.offset.samples.min()) and int(ds.offset.samples.max()) <= 92

    m = models["cosmos"](S=1, K=2, device="cuda", dtype="float")
    m.load(tmp_path)
    m.init(lr=0.005, nbatch_size=N, fbatch_size=F)
    m.run(2500, progress_bar=lambda x: x)
    z = m.z_probs[: N // 2, :, 0, 1].cpu() > 0.5
    truth = torch.as_tensor(sim.labels["z"][:, :, 0]).bool()
    tp, tn = float((z & truth).sum()), float((~z & ~truth).sum())
    fp, fn = float((z & ~truth).sum()), float((~z & truth).sum())
    assert (tp * tn - fp * fn) / math.sqrt((tp + fp) * (tp + fn) * (tn + fp) * (tn + fn)) > 0.9


@pytest.mark.gpu
def test_read_glimpse_more_aois_than_one_launch_covers(tmp_path):
    """70 000 AOIs: tq_glimpse_extract takes at most 65 535 per call (grid.y), the host splits the rest off."""
    cfg, _ = write_experiment(tmp_path / "raw", F=2, n_on=40000, n_off=30000, P=6, labels=False, aoiinfo_frame=1, seed=9)
    ds = read_glimpse(tmp_path, None, **cfg)
    want = og.read_glimpse(**cfg)
    assert ds.images.shape == (70000, 2, 1, 6, 6)
    assert torch.equal(ds.images, want["images"]) and torch.equal(ds.xy, want["xy"])
    assert torch.equal(ds.offset.samples.cpu(), want["offset_samples"]) and torch.equal(ds.offset.weights.cpu(), want["offset_weights"])


def golden_case():
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_glimpse_golden.py")
    spec = importlib.util.spec_from_file_location("make_glimpse_golden", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, np.load(os.path.join(os.path.dirname(path), "glimpse_golden.npz"))


def test_oracle_reproduces_the_committed_golden_output():
    mod, gold = golden_case()
    now = mod.build()
    for key in gold.files:
        assert np.array_equal(now[key], gold[key]), key


@pytest.mark.gpu
def test_read_glimpse_matches_the_committed_golden_output(tmp_path):
    mod, gold = golden_case()
    cfg, _ = write_experiment(tmp_path / "raw", **mod.CASE)
    cfg["bin-size"] = 3
    ds = read_glimpse(tmp_path, None, **cfg)
    assert np.array_equal(ds.images.numpy(), gold["images"]) and np.array_equal(ds.xy.numpy(), gold["xy"])
    assert np.array_equal(ds.offset.samples.cpu().numpy(), gold["offset_samples"])
    assert np.array_equal(ds.offset.weights.cpu().numpy(), gold["offset_weights"])
    assert np.array_equal(ds.labels["z"], gold["labels_z"]) and np.array_equal(ds.ttb.numpy(), gold["ttb"])
