"""
Device-side state of a cosmos fit and the launcher of the HIP step.

Owns what the reference keeps in Pyro's global param store + ``pyro.optim.Adam`` state
(tapqir/models/model.py:153-186): the flat unconstrained parameter buffer, its gradient and the
two Adam moments, the dataset resident in HBM (the reference re-uploads every minibatch,
tapqir/utils/dataset.py:140-151) and the step workspace.  All arithmetic happens in
``libtapqir_hip.so`` (include/tapqir_hip.h); torch is used for allocation, streams and the
``torch.distributed`` all-reduce only.
"""

import ctypes as C
import math
import os

import torch

from tapqir_amd import _lib
from tapqir_amd.exceptions import HipExtensionError
from tapqir_amd.models.layout import ParamLayout

DEFAULT_PRIORS = {  # tapqir/models/cosmos.py:55-64
    "background_mean_std": 1000.0, "background_std_std": 100.0, "lamda_rate": 1.0, "height_std": 10000.0,
    "width_min": 0.75, "width_max": 2.25, "proximity_rate": 1.0, "gain_std": 50.0,
}


def merge_offsets(samples, weights):
    """Offset samples with equal value are one mixture component: sum_o w_o f(D - d_o) is
    unchanged when equal d_o are merged (simulate.py:92-103 stores 3 identical samples)."""
    s = samples.detach().double().cpu()
    w = weights.detach().double().cpu()
    uniq, inv = torch.unique(s, return_inverse=True)
    wsum = torch.zeros_like(uniq).index_add_(0, inv, w)
    eps = torch.finfo(torch.float64).eps
    logits = torch.log(wsum.clamp(eps, 1 - eps)) if len(uniq) > 1 else torch.log(wsum.clamp(min=eps))
    return uniq.float(), logits.float()


class CosmosEngine:
    # what the library's launch sequence supports (tests/helpers.py derives a checker of the kernels' inline math that
    # runs the same bodies in plain host loops and switches these off)
    pipelined_tail = True      # tq_cosmos_step_overlapped: the tail of step t inside the sampling launch of step t+1
    split_sampling = True      # tq_cosmos_sample_locals_range: sampling split around an in-flight all-reduce
    lazy_adam_default = True   # tq_cosmos_adam_catchup

    def __init__(self, data, K=2, priors=None, device="cuda", eps=None, seed=0, n_offset=0, Nt_global=None,
                 crosstalk=False, hbm_budget=None):
        """``hbm_budget`` (bytes; default: the TAPQIR_AMD_HBM_BUDGET environment variable, else unlimited): device memory the
        IMAGES may take.  A data set whose resident form (the images and their interleaved copy) is larger is STREAMED: the
        images stay in page-locked host memory and every step brings the AOIs it needs through two device windows, group by
        group (the reference fetches every minibatch from host memory, tapqir/utils/dataset.py:140-151); parameters, optimiser
        state and the 12 B per unit of data statistics stay resident (section "streamed data sets" of DESIGN.md)."""
        self.device = torch.device(device)
        self.lib = self._open_library()
        lib = self.lib
        # With one dye and one channel the crosstalk model IS cosmos (alpha = [[1]] is a one-component Dirichlet:
        # constant draw, zero log-density, no gradient; the reference's smoke test runs this case,
        # test/test_tapqir.py:27-30): the cosmos kernels run and alpha_mean / alpha_size are inert parameters.
        nch = int(data.images.shape[2])
        self.crosstalk = bool(crosstalk) and nch > 1
        self.layout_crosstalk = bool(crosstalk)
        if self.crosstalk and (nch != 2 or int(K) > 2):
            raise ValueError("the crosstalk model is implemented for Q = C <= 2 dyes/channels and K <= 2 "
                             "(2^(K Q) joint spot-presence combinations; tapqir/models/crosstalk.py is experimental "
                             "upstream and indexes dyes and channels alike)")
        self.K = int(K)
        self.priors = dict(DEFAULT_PRIORS if priors is None else priors)
        self.eps = float(torch.finfo(torch.float32).eps if eps is None else eps)
        self.seed = int(seed)
        Nt, F, Cc, P = data.images.shape[:4]
        self.Nt, self.F, self.C, self.P = int(Nt), int(F), int(Cc), int(P)
        self.Nt_global = int(Nt_global) if Nt_global else self.Nt
        self.n_offset = int(n_offset)
        dev = self.device
        f32 = torch.float32
        if hbm_budget is None and os.environ.get("TAPQIR_AMD_HBM_BUDGET"):
            hbm_budget = int(float(os.environ["TAPQIR_AMD_HBM_BUDGET"]))
        img_bytes = 4 * data.images.numel()
        self.streamed = bool(hbm_budget) and 2 * img_bytes > hbm_budget and self.device.type == "cuda"
        if self.streamed:
            if self.crosstalk:
                raise ValueError("streaming of data sets larger than the device memory is implemented for the cosmos model")
            self._init_windows(data, int(hbm_budget))
        else:
            self.images = data.images.to(dev, f32).contiguous()
        self.xy = data.xy.to(dev, f32).contiguous()
        self.is_ontarget = data.is_ontarget.to(dev, torch.uint8).contiguous()
        self.mask = data.mask.to(dev, torch.uint8).contiguous()
        # the kernels take NULL for "no AOI is masked" (the default, dataset.py:63-65) and then skip the per-unit mask load
        self._mask_arg = self.mask if not bool(data.mask.all()) else None
        # tile-interleaved copy for the contiguous-batch pixel kernel (include/tapqir_hip.h); built by the
        # library so that any C caller gets the same layout
        self.images_il = None if self.streamed else self._interleaved_images()
        off_s, off_l = merge_offsets(data.offset.samples, data.offset.weights)
        self.offset_samples, self.offset_logits = off_s.to(dev), off_l.to(dev)
        self.O = int(off_s.numel())
        # per-unit data statistics of the single-offset formulation (sum v, sum ln v, #masked pixels)
        self.pixstats = None
        if self.O == 1:
            U = self.Nt * self.F * self.C
            self.pixstats = torch.empty(3 * U, dtype=f32, device=dev)
            if self.streamed:
                self._image_stats_streamed()
            else:
                self._image_stats(U)
        self.layout = ParamLayout(self.Nt, self.F, self.C, self.K, self.P, self.eps, crosstalk=self.layout_crosstalk)
        n = self.layout.total
        self._params = torch.zeros(n, dtype=f32, device=dev)
        self.grad = torch.zeros(n, dtype=f32, device=dev)
        self._exp_avg = torch.zeros(n, dtype=f32, device=dev)
        self._exp_avg_sq = torch.zeros(n, dtype=f32, device=dev)
        # lazy Adam of minibatch steps (include/tapqir_hip.h: tq_cosmos_adam_catchup): per-unit clock of the last update;
        # meaningful only while `_stale` (some unit's local parameters lag behind adam_step).  TAPQIR_AMD_LAZY_ADAM=0:
        # every minibatch step streams the whole buffers through the dense Adam kernel instead
        self.lazy_adam = self.lazy_adam_default and os.environ.get("TAPQIR_AMD_LAZY_ADAM", "1") != "0"
        self._last_step = torch.zeros(self.Nt * self.F * self.C, dtype=torch.int32, device=dev)
        self._stale = False
        gsz, bsz = self.struct_sizes()
        self.globals = torch.zeros(gsz // 4, dtype=f32, device=dev)
        self.gbase = torch.zeros(bsz // 8, dtype=torch.float64, device=dev)
        self._gsum_buf = torch.zeros(32, dtype=torch.float64, device=dev)  # TQ_GSUM_LEN
        self.n_gsum = 3 + 3 * self.C + (self.C * self.C if self.crosstalk else 0)
        self.gsum = self._gsum_buf[: self.n_gsum]  # the part that crosses ranks
        self.elbo_out = torch.zeros(1, dtype=torch.float64, device=dev)
        self._ws_key = None
        self._pending = None  # (args, handle) of a step whose all-reduce is in flight
        self.adam_step = 0
        self.lr, self.betas, self.adam_eps = 0.005, (0.9, 0.999), 1e-8
        # contiguous batches at least this large use the lane-per-unit pixel kernel (64 units per wave):
        # 65536 units = one wave per SIMD of an MI355X
        self.il_min_units = 65536
        # full-batch single-GPU steps leave their single-workgroup tail pending and run it inside the next step's
        # sampling launch (include/tapqir_hip.h: tq_cosmos_step_overlapped); TAPQIR_AMD_OVERLAP=0 turns it off
        self.overlap_tail = os.environ.get("TAPQIR_AMD_OVERLAP", "1") != "0"
        self._tail_args = None  # arguments of the step whose tail is pending
        # form of the backward pixel kernel for contiguous batches (include/tapqir_hip.h: pixel_mode): None = not chosen
        # yet -> the first full-batch step times both forms on this box (autotune_pixel)
        self.pixel_mode = None
        # full-batch single-GPU steps can run the pixel kernel and the per-unit kernel as ONE launch (pixel_mode =
        # TQ_PIXEL_FUSED_UNIT of tq_cosmos_args): None = not chosen yet -> the first such step times both ways
        # (autotune_fused); TAPQIR_AMD_FUSE_UNIT=0/1 fixes the choice
        # Default: fused wherever the step qualifies (_fusable).  Rounds 1-2 timed both ways on the first step, but four steps
        # each decide by ~2 % of noise while the fused launch has been 7-15 % ahead on every box since its per-unit phase
        # became scratch-free (round 3: a mis-tuned run of bench.py measured 0.244 instead of 0.227 ms per c2 step).
        # TAPQIR_AMD_FUSE_UNIT=auto brings the timing back (three interleaved rounds, best of each).
        fu = os.environ.get("TAPQIR_AMD_FUSE_UNIT", "1")
        self.fuse_unit = None if fu == "auto" else fu != "0"
        # minibatch steps with the lazy Adam clock run as ONE launch (include/tapqir_hip.h: tq_cosmos_minibatch_step);
        # TAPQIR_AMD_MB_FUSED=0 keeps the five-launch sequence
        self.fused_minibatch = self.pipelined_tail and os.environ.get("TAPQIR_AMD_MB_FUSED", "1") != "0"
        self._sync = torch.zeros(64, dtype=torch.int32, device=dev)  # TQ_SYNC_WORDS (tickets, flags; diagnostic stamps of a TQ_MB_STAMPS build)
        self._sync_value = 0

    # -- the library --------------------------------------------------------------------------------
    def _open_library(self):
        if self.device.type != "cuda":
            raise HipExtensionError(
                "the cosmos SVI step only runs on an AMD GPU through libtapqir_hip.so "
                f"(device={str(self.device)!r} requested); there is no CPU path")
        return _lib.load()

    def struct_sizes(self):
        """(bytes of TqGlobals, bytes of TqGlobalBase)."""
        return int(self.lib.tq_globals_size()), int(self.lib.tq_gbase_size())

    def _interleaved_images(self):
        """Tile-interleaved copy for the contiguous-batch pixel kernel (include/tapqir_hip.h); built by the library so
        that any C caller gets the same layout.  cosmos: one (P, P) tile per unit (n, f, c); crosstalk: one (C, P, P)
        tile per AOI-frame (n, f)."""
        tiles = self.Nt * self.F * (1 if self.crosstalk else self.C)
        npix = self.P * self.P * (self.C if self.crosstalk else 1)
        out = torch.empty(int(self.lib.tq_interleaved_floats_n(tiles, npix)), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.tq_images_interleave_n(_lib.ptr(self.images), _lib.ptr(out), tiles, npix, self._stream()),
                   "tq_images_interleave_n")
        return out

    def _image_stats(self, U):
        _lib.check(self.lib.tq_image_stats(_lib.ptr(self.images), _lib.ptr(self.offset_samples), _lib.ptr(self.pixstats),
                                           U, self.P, self._stream()), "tq_image_stats")

    def _blk_floats(self, B):
        return int(self.lib.tq_cosmos_blk_floats(self.Nt, self.F, self.C, int(self.crosstalk), B))

    def _run_snr_chi2(self, a):
        _lib.check(self.lib.tq_snr_chi2(C.byref(a), self._stream()), "tq_snr_chi2")

    def snr_chi2(self, offset_mean, offset_var):
        """Signal-to-noise ratio of every spot ``(K, Nt, F, Q)`` and chi2 of the fitted image ``(Nt, F, Q)`` at the
        posterior means of the variational parameters (tapqir/utils/stats.py:29-86, 166-182), computed where the images
        are: one kernel over all units instead of the reference's host loop over AOIs."""
        cp = self.layout.constrained(self.params, {"h_loc", "w_mean", "x_mean", "y_mean", "b_loc", "gain_loc"})
        f32 = torch.float32
        U = self.Nt * self.F * self.C
        flat = lambda t: t.detach().to(f32).reshape(self.K, U).contiguous()
        h, w, x, y = flat(cp["h_loc"]), flat(cp["w_mean"]), flat(cp["x_mean"]), flat(cp["y_mean"])
        b = cp["b_loc"].detach().to(f32).reshape(U).contiguous()
        snr = torch.empty(self.K, U, dtype=f32, device=self.device)
        chi2 = torch.empty(U, dtype=f32, device=self.device)
        gain = float(cp["gain_loc"])
        p = _lib.ptr

        def launch(images, xy, hh, ww, xx, yy, bb, snr_o, chi2_o, n_units):
            a = _lib.SnrArgs()
            a.images, a.xy = p(images), p(xy)
            a.height, a.width, a.x, a.y, a.background = p(hh), p(ww), p(xx), p(yy), p(bb)
            a.snr, a.chi2 = p(snr_o), p(chi2_o)
            a.U, a.P, a.K = n_units, self.P, self.K
            a.gain, a.offset_mean, a.offset_var = gain, float(offset_mean), float(offset_var)
            self._run_snr_chi2(a)

        if not self.streamed:
            launch(self.images, self.xy, h, w, x, y, b, snr, chi2, U)
        else:  # window by window (whole AOIs): the same kernel on slices of every array
            fc = self.F * self.C
            xy2 = self.xy.reshape(U, 2)
            for aois in self._groups(None):
                wi, n0, n = self._upload_group(aois), int(aois[0]), aois.numel()
                torch.cuda.current_stream(self.device).wait_event(self._win_ready[wi])
                u0, u1 = n0 * fc, (n0 + n) * fc
                parts = [t[:, u0:u1].contiguous() for t in (h, w, x, y)]
                snr_g = torch.empty(self.K, u1 - u0, dtype=f32, device=self.device)
                chi2_g = torch.empty(u1 - u0, dtype=f32, device=self.device)
                launch(self._win[wi], xy2[u0:u1].contiguous(), *parts, b[u0:u1].contiguous(), snr_g, chi2_g, u1 - u0)
                snr[:, u0:u1] = snr_g
                chi2[u0:u1] = chi2_g
                ev = torch.cuda.Event()
                ev.record()
                self._win_free[wi] = ev
        return snr.view(self.K, self.Nt, self.F, self.C), chi2.view(self.Nt, self.F, self.C)

    def run_probs(self, a):
        _lib.check(self.lib.tq_cosmos_probs(C.byref(a), self._stream()), "tq_cosmos_probs")

    # -- streamed data sets (images larger than the device memory they may take) --------------------------
    def _init_windows(self, data, budget):
        """Images in page-locked host memory; two device windows of W whole AOIs each (all frames of an AOI travel together:
        the per-AOI sites need every frame of the batch, and an AOI's F C P P floats are contiguous on the host)."""
        f32 = torch.float32
        self.images = None
        self.images_host = data.images.to("cpu", f32).contiguous().pin_memory()
        Nt = self.images_host.shape[0]
        per_aoi = 4 * self.images_host[0].numel()
        self.window_aois = int(max(1, min(Nt, budget // (2 * per_aoi))))
        shape = (self.window_aois,) + tuple(self.images_host.shape[1:])
        self._win = [torch.empty(shape, dtype=f32, device=self.device) for _ in range(2)]
        self._win_free = [None, None]  # event after the last launch that reads the window
        self._win_ready = [torch.cuda.Event(), torch.cuda.Event()]
        self._copy_stream = torch.cuda.Stream(self.device)
        self._win_count = 0

    def _upload_group(self, aois):
        """Bring the AOIs `aois` (host int64 tensor, in batch order) into the next window on the copy stream; returns the
        window index.  Runs of consecutive AOIs are one copy."""
        w = self._win_count % 2
        self._win_count += 1
        win = self._win[w]
        with torch.cuda.stream(self._copy_stream):
            if self._win_free[w] is not None:
                self._copy_stream.wait_event(self._win_free[w])  # the kernels that read this window two groups ago are done
            idx = aois.tolist()
            j = 0
            while j < len(idx):
                k = j
                while k + 1 < len(idx) and idx[k + 1] == idx[k] + 1:
                    k += 1
                win[j:k + 1].copy_(self.images_host[idx[j]:idx[k] + 1], non_blocking=True)
                j = k + 1
            self._win_ready[w].record(self._copy_stream)
        return w

    def _groups(self, ndx):
        aois = torch.arange(self.Nt) if ndx is None else ndx.detach().to("cpu", torch.int64).reshape(-1)
        return [aois[i:i + self.window_aois] for i in range(0, aois.numel(), self.window_aois)]

    def _image_stats_streamed(self):
        U, fc = self.Nt * self.F * self.C, self.F * self.C
        groups = self._groups(None)
        nxt = self._upload_group(groups[0])
        for g, aois in enumerate(groups):
            w, n0, n = nxt, int(aois[0]), aois.numel()
            torch.cuda.current_stream(self.device).wait_event(self._win_ready[w])
            if g + 1 < len(groups):
                nxt = self._upload_group(groups[g + 1])
            tmp = torch.empty(3, n * fc, dtype=torch.float32, device=self.device)
            _lib.check(self.lib.tq_image_stats(_lib.ptr(self._win[w]), _lib.ptr(self.offset_samples), _lib.ptr(tmp), n * fc, self.P,
                                               self._stream()), "tq_image_stats")
            self.pixstats.view(3, U)[:, n0 * fc:(n0 + n) * fc] = tmp
            ev = torch.cuda.Event()
            ev.record()
            self._win_free[w] = ev

    def _step_streamed(self, ndx, fdx):
        """One SVI step of a streamed data set: the step's AOIs in groups of at most `window_aois`; per group the staged
        launches of a gathered batch (lazy-Adam catch-up for minibatches, local sampling, likelihood + per-unit + per-AOI
        terms with the Adam of the local parameters fused in) on the window the copy stream has filled meanwhile; the
        cross-unit sums of the groups are added and ONE tail (global sites, ELBO, Adam of the per-AOI / global parameters)
        closes the step.  Draws are keyed by the global unit index, so the step is the resident step up to the order in
        which the groups' sums are added."""
        self._finish_pending()
        self._finish_tail()
        groups = self._groups(ndx)
        nb_total = sum(g.numel() for g in groups)
        fb = self.F if fdx is None else int(fdx.numel())
        minibatch = nb_total < self.Nt or fb < self.F
        lazy = minibatch and self.lazy_adam
        self._workspace(min(self.window_aois, nb_total), fb)  # (a new workspace joins first: before the lazy clock is started)
        ws_key = self._ws_key
        if lazy:
            if not self._stale:
                self._last_step.fill_(self.adam_step)
                self._stale = True
        else:
            self._catch_up_all()
        fdx_dev = None if fdx is None else self._index_to_device(fdx, 1)
        gtot = torch.zeros_like(self.gsum)
        cur = torch.cuda.current_stream(self.device)
        nxt = self._upload_group(groups[0])
        a = None
        for g, aois in enumerate(groups):
            w = nxt
            nd = aois.to(self.device, torch.int32)
            a = self.make_args(nd, fdx_dev, draw_globals=True, _for_step=True, _ws=ws_key)
            a.images, a.images_il, a.images_by_slot = _lib.ptr(self._win[w]), None, 1
            a.scale_n = self.Nt_global / self._nb_global(nb_total)
            a.scale = a.scale_n * self.F / fb
            a.zero_grad = int(minibatch)
            a.fuse_adam = int(not minibatch or lazy)
            if not lazy:
                a.last_step = None
            if g == 0:
                self.call("cosmos_sample_globals", a)
            cur.wait_event(self._win_ready[w])
            if g + 1 < len(groups):
                nxt = self._upload_group(groups[g + 1])  # travels while this group computes
            if lazy:
                self._adam_catchup(a, 0)
            self.call("cosmos_sample_locals", a)
            self.call("cosmos_elbo_grads", a)
            gtot += self.gsum
            ev = torch.cuda.Event()
            ev.record(cur)
            self._win_free[w] = ev
        self.gsum.copy_(gtot)
        self._tail_reduced(a, None)
        self.adam_step += 1

    # -- parameter / moment buffers ---------------------------------------------------------------
    # Reading them from outside the step completes deferred work first: the pending tail of a pipelined step and the
    # zero-gradient Adam steps that lazy minibatch steps still owe to the units outside their minibatches.
    @property
    def params(self):
        self.join()
        return self._params

    @property
    def exp_avg(self):
        self.join()
        return self._exp_avg

    @property
    def exp_avg_sq(self):
        self.join()
        return self._exp_avg_sq

    def _catch_up_all(self):
        """Bring the local parameters and moments of every unit to `adam_step` (lazy Adam)."""
        if self._stale:
            self._stale = False
            a = _lib.CosmosArgs()  # only what the replay reads: buffers, geometry, optimiser constants, step count
            p = _lib.ptr
            a.params, a.exp_avg, a.exp_avg_sq, a.last_step = p(self._params), p(self._exp_avg), p(self._exp_avg_sq), p(self._last_step)
            a.globals, a.gbase = p(self.globals), p(self.gbase)
            a.Nt, a.F, a.C, a.P, a.K, a.O = self.Nt, self.F, self.C, self.P, self.K, self.O
            a.nb, a.fb = self.Nt, self.F
            a.lr, a.beta1, a.beta2, a.adam_eps = self.lr, self.betas[0], self.betas[1], self.adam_eps
            a.beta1_d, a.beta2_d = float(self.betas[0]), float(self.betas[1])
            a.crosstalk = int(self.crosstalk)
            a.step = self.adam_step
            self._adam_catchup(a, 1)

    def _adam_catchup(self, a, all_units):
        _lib.check(self.lib.tq_cosmos_adam_catchup(C.byref(a), all_units, self._stream()), "tq_cosmos_adam_catchup")

    def reset_adam_clock(self, step=0):
        """The buffers were (re)written from outside (initialisation, checkpoint): everything is current at `step`."""
        self.join()
        self.adam_step = int(step)
        self._stale = False
        # the ticket / flag words of the single-launch minibatch step start from a known state again: a launch that timed
        # out waiting for its flag (the kernel then leaves a NaN loss, which brings Model.run here through its checkpoint
        # reload) must not leave a ticket counter that no later launch can start from
        self._sync.zero_()
        self._sync_value = 0
        if "_sub" in self.__dict__:
            self._sub["ready"] = None  # a device-drawn subsample belongs to the step count it was drawn for

    # -- workspace ---------------------------------------------------------------------------------
    def _workspace(self, nb, fb):
        key = (nb, fb)
        if self._ws_key == key:
            return
        self.join()
        K, M = self.K, 1 << self.K
        B = nb * fb * self.C
        dev, f32 = self.device, torch.float32
        self.lat = torch.zeros((1 + 4 * K) * B, dtype=f32, device=dev)
        self.site = torch.zeros(int(os.environ.get("TAPQIR_AMD_SITE_ROWS", 5)) * (1 + 4 * K) * B, dtype=f32, device=dev)  # TQ_NSITE_STORED rows (the switch: A/B runs against older builds of the library)
        self.pix = torch.zeros((M + 2 + 4 * K + (1 + self.C if self.crosstalk else 0)) * B, dtype=f32, device=dev)
        self.aoi_part = torch.zeros(3 * B, dtype=f32, device=dev)
        nblk = (B + 255) // 256
        self.blk_part = torch.zeros(self._blk_floats(B), dtype=f32, device=dev)
        self._ws_key = key

    def _index_to_device(self, idx, which):
        """One subsample index vector as a device int32 tensor (see _indices_to_device for the step's fast path)."""
        return idx.to(self.device, torch.int32).contiguous()

    def _indices_to_device(self, ndx, fdx):
        """Subsample indices of a step as device int32 tensors.  Host tensors (what ``torch.randperm`` gives the caller
        every step) take ONE asynchronous copy: both vectors are written, converted to int32, into a pinned staging
        slot and copied into a persistent device slot of the same ring -- no allocation, no pageable copy (which would
        block the host until the stream has drained) and a few microseconds of host time instead of ~40 per vector,
        which made the host the bottleneck of the default 10 x 512 minibatch step."""
        host = [t is not None and t.device.type == "cpu" for t in (ndx, fdx)]
        if self.device.type != "cuda" or not any(host):
            return (None if ndx is None else self._index_to_device(ndx, 0)), (None if fdx is None else self._index_to_device(fdx, 1))
        pin, pin_np, dev, ev, _ = slot = self._idx_slot()
        pos, out = 0, []
        for t, h in zip((ndx, fdx), host):
            if t is None:
                out.append(None)
            elif h:
                k = t.numel()
                pin_np[pos:pos + k] = t.numpy().reshape(-1)  # int64 -> int32
                out.append((pos, k))
                pos += k
            else:
                out.append(t.to(self.device, torch.int32).contiguous())
        dev[:pos].copy_(pin[:pos], non_blocking=True)
        ev.record()  # on the current stream
        slot[4] = True
        return tuple(dev[o[0]:o[0] + o[1]] if isinstance(o, tuple) else o for o in out)

    def _idx_slot(self):
        """Next slot of the 16-deep ring of pinned + device index buffers ([Nt | F] int32 each)."""
        ring = self.__dict__.get("_idx_ring")
        if ring is None:
            n = self.Nt + self.F
            ring = []
            for _ in range(16):
                pin = torch.empty(n, dtype=torch.int32).pin_memory()
                ring.append([pin, pin.numpy(), torch.empty(n, dtype=torch.int32, device=self.device), torch.cuda.Event(), False])
            self._idx_ring, self._idx_count = ring, 0
        slot = ring[self._idx_count % len(ring)]
        self._idx_count += 1
        if slot[4]:
            slot[3].synchronize()  # the copy that last read this staging slot (16 steps ago) has completed
        return slot

    def draw_subsample(self, nb, fb, generator):
        """``torch.randperm(Nt)[:nb]``, ``torch.randperm(F)[:fb]`` (pyro.plate's subsample, cosmos.py:194-208) drawn straight
        into a pinned staging slot as int32 and copied to the device in ONE asynchronous copy: returns device index
        tensors for ``step`` (None where the whole axis is taken).  Saves the slicing and int64 -> int32 copies of
        ``_indices_to_device`` -- at the default minibatch the host, not the device, bounds ``Model.run``."""
        if self.device.type != "cuda":
            raise RuntimeError("draw_subsample stages through pinned memory: GPU engines only")
        take_n, take_f = nb < self.Nt, fb < self.F
        if not (take_n or take_f):
            return None, None
        pin, _, dev, ev, _ = slot = self._idx_slot()
        Nt, F = self.Nt, self.F
        if take_n:
            torch.randperm(Nt, generator=generator, dtype=torch.int32, out=pin[:Nt])
        if take_f:
            torch.randperm(F, generator=generator, dtype=torch.int32, out=pin[Nt:Nt + F])
        if take_n and take_f:
            dev.copy_(pin, non_blocking=True)
        elif take_n:
            dev[:nb].copy_(pin[:nb], non_blocking=True)
        else:
            dev[Nt:Nt + fb].copy_(pin[Nt:Nt + fb], non_blocking=True)
        ev.record()
        slot[4] = True
        return (dev[:nb] if take_n else None), (dev[Nt:Nt + fb] if take_f else None)

    def make_args(self, ndx=None, fdx=None, draw_globals=True, global_weight=1.0, step=None, draw_locals=None,
                  _for_step=False, _ws=None):
        if not _for_step:
            # staged calls from outside read the parameters of arbitrary units and overwrite globals / gsum / grad: finish
            # a pipelined tail or an in-flight all-reduce first, then bring every unit to the current Adam step
            self.join()
        nb = self.Nt if ndx is None else int(ndx.numel())
        fb = self.F if fdx is None else int(fdx.numel())
        if _ws is None:
            self._workspace(nb, fb)  # (_ws: the caller has set up a workspace for batches up to that size)
        ndx, fdx = self._indices_to_device(ndx, fdx)
        # keep the index tensors alive while kernels run (the previous step's too: its tail may still be pending)
        self._keep_prev, self._keep = getattr(self, "_keep", None), (ndx, fdx)
        p = _lib.ptr
        a = _lib.CosmosArgs()
        a.images, a.xy, a.is_ontarget, a.aoi_mask = p(self.images), p(self.xy), p(self.is_ontarget), p(self._mask_arg)
        a.images_il = p(self.images_il)
        a.pixstats = p(self.pixstats)
        a.ndx, a.fdx = p(ndx), p(fdx)
        a.offset_samples, a.offset_logits = p(self.offset_samples), p(self.offset_logits)
        a.params, a.grad, a.exp_avg, a.exp_avg_sq = p(self._params), p(self.grad), p(self._exp_avg), p(self._exp_avg_sq)
        a.last_step = p(self._last_step)
        a.beta1_d, a.beta2_d = float(self.betas[0]), float(self.betas[1])
        a.lat, a.pix, a.aoi_part, a.blk_part = p(self.lat), p(self.pix), p(self.aoi_part), p(self.blk_part)
        a.site = p(self.site)
        a.draw_locals = int(bool(draw_globals if draw_locals is None else draw_locals))
        a.il_min_units = self.il_min_units
        a.gsum, a.globals, a.gbase, a.elbo_out = p(self._gsum_buf), p(self.globals), p(self.gbase), p(self.elbo_out)
        a.Nt, a.F, a.C, a.P, a.K, a.O = self.Nt, self.F, self.C, self.P, self.K, self.O
        a.nb, a.fb, a.n_offset, a.draw_globals = nb, fb, self.n_offset, int(bool(draw_globals))
        a.scale_n = self.Nt_global / self._nb_global(nb)
        a.scale = a.scale_n * self.F / fb
        a.global_weight = global_weight
        a.eps = self.eps
        pr = self.priors
        a.width_min, a.width_max, a.height_std = pr["width_min"], pr["width_max"], pr["height_std"]
        a.background_mean_std, a.background_std_std = pr["background_mean_std"], pr["background_std_std"]
        a.gain_std, a.lamda_rate, a.proximity_rate = pr["gain_std"], pr["lamda_rate"], pr["proximity_rate"]
        t = self.adam_step + 1
        a.lr, a.beta1, a.beta2, a.adam_eps = self.lr, self.betas[0], self.betas[1], self.adam_eps
        a.bias_correction1 = 1.0 - self.betas[0] ** t
        a.bias_correction2 = 1.0 - self.betas[1] ** t
        a.zero_grad = int(nb < self.Nt or fb < self.F)
        a.fuse_adam = 0  # set by step() for full-batch steps
        a.crosstalk = int(self.crosstalk)
        a.seed = self.seed
        a.step = self.adam_step if step is None else int(step)
        a.pixel_mode = int(self.pixel_mode or 0)
        a.sync, a.tail_kind = p(self._sync), 0
        return a

    def _step_args(self, ndx, fdx):
        """make_args for step(): the ~60 fields are filled once per batch geometry / optimiser setting and copied
        afterwards; only the subsample pointers, the step count and the bias corrections change from step to step
        (minibatch steps are launch-bound: the host side of a step must stay well below the ~50 us of its kernels)."""
        nb = self.Nt if ndx is None else int(ndx.numel())
        fb = self.F if fdx is None else int(fdx.numel())
        key = (nb, fb, self.lr, self.betas, self.adam_eps, self.seed, id(self.priors), self._ws_key)
        if self.__dict__.get("_tmpl_key") != key or self._ws_key != (nb, fb):
            a = self.make_args(ndx, fdx, _for_step=True)
            self._tmpl, self._tmpl_key = _lib.CosmosArgs.from_buffer_copy(a), (nb, fb, self.lr, self.betas, self.adam_eps,
                                                                              self.seed, id(self.priors), self._ws_key)
            return a
        a = _lib.CosmosArgs.from_buffer_copy(self._tmpl)
        ndx, fdx = self._indices_to_device(ndx, fdx)
        self._keep_prev, self._keep = self._keep, (ndx, fdx)
        a.ndx, a.fdx = _lib.ptr(ndx), _lib.ptr(fdx)
        t = self.adam_step + 1
        a.bias_correction1 = 1.0 - self.betas[0] ** t
        a.bias_correction2 = 1.0 - self.betas[1] ** t
        a.step = self.adam_step
        return a

    def _nb_global(self, nb):
        # AOI sharding: every rank subsamples nb local AOIs of Nt local ones, so the global
        # subsample is nb * (Nt_global / Nt)
        return nb * self.Nt_global / self.Nt

    # -- launches ------------------------------------------------------------------------------------
    def _stream(self):
        # the raw handle of torch's current stream (torch.cuda.current_stream() builds a Stream object: ~9 us a call,
        # several calls per step)
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(index))

    def call(self, name, args):
        _lib.check(getattr(self.lib, "tq_" + name)(C.byref(args), self._stream()), "tq_" + name)

    def ksmogn_args(self, backward=True):
        """Argument block of ``tq_ksmogn_log_prob`` for the full batch with the latents in the workspace (diagnostics,
        bench.py's roofline leg and the autotuner)."""
        self._workspace(self.Nt, self.F)
        K, M = self.K, 1 << self.K
        B = self.Nt * self.F * self.C
        k = _lib.KsmognArgs()
        p = _lib.ptr
        k.images, k.images_il, k.xy, k.ndx, k.fdx = p(self.images), p(self.images_il), p(self.xy), None, None
        k.nb_full, k.il_min_units = self.Nt, self.il_min_units
        k.pixstats, k.stats_stride = p(self.pixstats), B
        f = lambda row: self.lat.data_ptr() + 4 * row * B
        k.background, k.height, k.width, k.x, k.y = f(0), f(1), f(1 + K), f(1 + 2 * K), f(1 + 3 * K)
        k.gain = self.globals.data_ptr()
        k.offset_samples, k.offset_logits = p(self.offset_samples), p(self.offset_logits)
        k.gout, k.m_logit, k.aoi_mask = None, p(self._params), p(self._mask_arg)
        g = lambda row: self.pix.data_ptr() + 4 * row * B
        k.ll = g(0)
        if backward:
            k.g_background, k.g_gain = g(M), g(M + 1)
            k.g_height, k.g_width, k.g_x, k.g_y = g(M + 2), g(M + 2 + K), g(M + 2 + 2 * K), g(M + 2 + 3 * K)
        k.m_kstride = B
        k.nb, k.fb, k.C, k.F, k.P, k.K, k.O = self.Nt, self.F, self.C, self.F, self.P, K, self.O
        k.scale = 1.0
        k.pixel_mode = int(self.pixel_mode or 0)
        return k

    def autotune_pixel(self, launches=8):
        """Choose the form of the backward pixel kernel for this box and this dataset: the persistent form is insensitive to
        wave-launch and memory latency, the one-wave-per-tile form is ahead where those are short (120-131 us against
        114-151 us at 400 000 units across the boxes of one pool).  Both give the same results; a few launches of each on
        the latents of the workspace (drawn here if the step has not run yet) decide.  Applies to K <= 2 with one offset."""
        self.pixel_mode = 0
        if self.crosstalk or self.K > 2 or self.O != 1 or self.P not in (14, 20) or self.Nt * self.F * self.C < self.il_min_units:
            return 0
        a = self.make_args()
        self.call("cosmos_sample_globals", a)
        self.call("cosmos_sample_locals", a)
        times = []
        for mode in (0, 1):
            self.pixel_mode = mode
            k = self.ksmogn_args(backward=True)
            for _ in range(2):
                _lib.check(self.lib.tq_ksmogn_log_prob(C.byref(k), self._stream()), "tq_ksmogn_log_prob")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(launches):
                self.lib.tq_ksmogn_log_prob(C.byref(k), self._stream())
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1) / launches)
        self.pixel_mode = int(times[1] < times[0])
        self.pixel_times_ms = times
        self.__dict__.pop("_tmpl_key", None)  # argument templates carry the mode
        return self.pixel_mode

    def _fusable(self):
        """Can a full-batch step of this engine run pixel + per-unit kernel in one launch?  (tq_cosmos.hip:
        tq_fused_pixel_unit -- same conditions.)"""
        return (self.pipelined_tail and not self.crosstalk and self.K <= 2 and self.O == 1 and self.P in (14, 20)
                and self.Nt * self.F * self.C >= self.il_min_units and self.F * self.C >= 256
                and os.environ.get("TAPQIR_AMD_ROWS", "1") != "0")

    def autotune_fused(self, steps=6, rounds=3):
        """Choose between two launches (pixel kernel, per-unit kernel) and the fused launch for the full-batch steps of
        this box and dataset by timing real steps each way (``rounds`` interleaved rounds of ``steps`` steps, the best round
        of each counts); parameters, optimiser state and step count are put back afterwards."""
        if not self._fusable():
            self.fuse_unit = False
            return False
        self.join()
        saved = (self.params.clone(), self.exp_avg.clone(), self.exp_avg_sq.clone(), self.adam_step)
        times = [float("inf"), float("inf")]
        for _ in range(rounds):
            for k, fuse in enumerate((False, True)):
                self.fuse_unit = fuse
                self.step()
                self.step()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(steps):
                    self.step()
                e1.record()
                e1.synchronize()
                times[k] = min(times[k], e0.elapsed_time(e1) / steps)
        self.join()
        self.params.copy_(saved[0])
        self.exp_avg.copy_(saved[1])
        self.exp_avg_sq.copy_(saved[2])
        self.adam_step = saved[3]
        self.fuse_unit = bool(times[1] < times[0])
        self.step_times_ms = times
        return self.fuse_unit

    def step_subsampled(self, nb, fb, generator=None):
        """A minibatch step on a subsample the DEVICE has drawn: the single-launch minibatch step also draws the next step's
        `randperm(Nt)[:nb]`, `randperm(F)[:fb]` (include/tapqir_hip.h: next_ndx / next_fdx) into the other of two index
        slots, so the host of a running fit neither draws, stages nor copies an index -- at the default 10 x 512 the two
        host-side randperm calls and their copy, not the GPU, bounded ``Model.run``.  The first step of a sequence (and any
        step whose batch sizes differ from what the previous launch prepared) takes a host draw from ``generator``.  Returns
        False if this engine cannot run that path (the caller then passes its own subsample to ``step``)."""
        nb, fb = min(int(nb), self.Nt), min(int(fb), self.F)
        if not (self.fused_minibatch and self.lazy_adam and self.pipelined_tail and not self.crosstalk and not self.streamed
                and (nb < self.Nt or fb < self.F) and fb * self.C >= 16 and max(self.Nt, self.F) <= 2048  # TQ_SUBSAMPLE_MAX
                and os.environ.get("TAPQIR_AMD_DEVICE_SUBSAMPLE", "1") != "0"):
            return False
        st = self.__dict__.get("_sub")
        if st is None:  # (not setdefault: its argument -- two device allocations + fills -- would be built at every step)
            st = self._sub = {"slots": [torch.zeros(self.Nt + self.F, dtype=torch.int32, device=self.device) for _ in range(2)],
                              "turn": 0, "ready": None}
        cur, Nt = st["turn"], self.Nt
        slot, nxt = st["slots"][cur], st["slots"][1 - cur]
        if st["ready"] != (nb, fb, self.adam_step):
            if nb < self.Nt:
                slot[:nb].copy_(torch.randperm(self.Nt, generator=generator)[:nb].to(torch.int32))
            if fb < self.F:
                slot[Nt:Nt + fb].copy_(torch.randperm(self.F, generator=generator)[:fb].to(torch.int32))
        ndx = slot[:nb] if nb < self.Nt else None
        fdx = slot[Nt:Nt + fb] if fb < self.F else None
        self.step(ndx, fdx, _next_sub=(nxt[:nb] if nb < self.Nt else None, nxt[Nt:Nt + fb] if fb < self.F else None))
        st["turn"], st["ready"] = 1 - cur, (nb, fb, self.adam_step)
        return True

    def step(self, ndx=None, fdx=None, allreduce=None, _next_sub=None):
        """One SVI step; returns nothing (the ELBO stays on the device in ``elbo_out``).

        ``allreduce(gsum)`` sums the cross-unit sums over the ranks of an AOI-sharded run.  If it returns a handle with
        ``wait()`` (an asynchronous collective) and the step is a full-batch one, the global tail of the step (global
        sites, Adam of the per-AOI and global parameters) is deferred: it runs after the NEXT step's local guide
        sampling, which needs local parameters only, so the collective's latency hides behind that kernel.  ``join()``
        (called by every read-out) completes a deferred tail."""
        if self.streamed:
            if allreduce is not None:
                raise NotImplementedError("AOI sharding of a streamed data set: shard first, each rank then streams its own AOIs "
                                          "(not built: a rank's shard of every BASELINE config fits its 288 GB many times over)")
            return self._step_streamed(ndx, fdx)
        if self.pixel_mode is None and ndx is None and fdx is None and self.pipelined_tail:
            self.autotune_pixel()
        if self.fuse_unit is None and ndx is None and fdx is None and allreduce is None and self.pipelined_tail:
            self.autotune_fused()
        a = self._step_args(ndx, fdx)
        minibatch = bool(a.zero_grad)
        # Adam on the local block is fused into the unit kernel: full batches, and minibatches with the lazy clock
        a.fuse_adam = int(not minibatch or self.lazy_adam)
        # (offset histograms stay in the single launch: the staged sequence with the wave-per-unit likelihood kernel measured
        # 0.150 against 0.162 ms at O = 50 -- TAPQIR_AMD_MB_HIST_STAGED=1 -- not worth a second path through the step)
        one_launch = (minibatch and self.lazy_adam and self.fused_minibatch and allreduce is None and not self.crosstalk
                      and a.fb * self.C >= 16 and not (self.O >= 8 and os.environ.get("TAPQIR_AMD_MB_HIST_STAGED") == "1"))
        # a SMALL full batch (BASELINE config c1: 5000 units) is the same case as a minibatch -- a few hundred workgroups
        # whose every phase is latency: the one-launch step with no subsample and no lazy clock instead of the two launches
        # of the pipelined full-batch step (TAPQIR_AMD_SMALL_FULL=0: the two launches)
        B_full = self.Nt * self.F * self.C
        if (not minibatch and self.fused_minibatch and self.pipelined_tail and allreduce is None and not self.crosstalk
                and B_full <= 10240 and B_full < self.il_min_units and self.F * self.C >= 16
                and os.environ.get("TAPQIR_AMD_SMALL_FULL", "1") != "0"):
            one_launch = True
        if minibatch and self.lazy_adam:
            if not self._stale:
                self._last_step.fill_(self.adam_step)  # every unit is current: start the clock here
                self._stale = True
            if not one_launch:
                self._adam_catchup(a, 0)
        else:
            self._catch_up_all()
            a.last_step = None  # full batch: no unit falls behind
        if allreduce is None and self.pipelined_tail:
            self._finish_pending()
            if one_launch:
                # catch-up, site draws, likelihood, per-unit terms + Adam of this step and the pending tail of the
                # previous one in a single launch; the tail of this step stays pending
                prev = self._tail_args
                self._sync_value = self._sync_value % 0x3FFFFFFF + 1  # never 0, never the same value twice in a row
                a.sync_value = self._sync_value
                if _next_sub is not None:
                    a.next_ndx, a.next_fdx = _lib.ptr(_next_sub[0]), _lib.ptr(_next_sub[1])
                _lib.check(self.lib.tq_cosmos_minibatch_step(C.byref(a), None if prev is None else C.byref(prev),
                                                              self._stream()), "tq_cosmos_minibatch_step")
                a.tail_kind = 1  # TQ_TAIL_ROWS16
                self._tail_args = a
            elif self.overlap_tail and a.fuse_adam:
                if self.fuse_unit and not minibatch and self._fusable():
                    a.pixel_mode = 2  # TQ_PIXEL_FUSED_UNIT
                prev = self._tail_args
                _lib.check(self.lib.tq_cosmos_step_overlapped(C.byref(a), None if prev is None else C.byref(prev),
                                                              self._stream()), "tq_cosmos_step_overlapped")
                self._tail_args = a
            else:
                self._finish_tail()
                self.call("cosmos_step", a)
        else:
            self._finish_tail()
            pending = self._pending
            if pending is not None and not a.fuse_adam:
                self._finish_pending()  # a minibatch step samples after the dense Adam of the previous one
                pending = None
            if pending is not None and self.split_sampling and getattr(pending[1], "in_stream", False):
                # the collective was issued on this stream (tapqir_amd.rccl.RcclDirect): stream order is the dependency.  ONE
                # sampling launch that carries the pending step's post-all-reduce tail and this step's global draws
                prev, _ = self._pending
                self._pending = None
                _lib.check(self.lib.tq_cosmos_sample_locals_range(C.byref(a), 0, 1 + 4 * self.K, C.byref(prev), self._stream()),
                           "tq_cosmos_sample_locals_range")
            elif pending is not None and self.split_sampling:
                # first half of the local sites while the all-reduce is in flight; the rest in a launch that also
                # carries the pending step's post-all-reduce tail and this step's global draws
                nsites = 1 + 4 * self.K
                n1 = nsites // 2
                prev, handle = self._pending
                self._pending = None
                _lib.check(self.lib.tq_cosmos_sample_locals_range(C.byref(a), 0, n1, None, self._stream()),
                           "tq_cosmos_sample_locals_range")
                handle.wait()  # the current stream waits for the collective
                _lib.check(self.lib.tq_cosmos_sample_locals_range(C.byref(a), n1, nsites - n1, C.byref(prev), self._stream()),
                           "tq_cosmos_sample_locals_range")
            else:
                self.call("cosmos_sample_locals", a)
                if pending is not None:
                    self._finish_pending(next_args=a)  # ... and draws this step's global sites in the same launch
                else:
                    self.call("cosmos_sample_globals", a)
            if a.fuse_adam and not minibatch and self.fuse_unit is not False and self._fusable():
                a.pixel_mode = 2  # TQ_PIXEL_FUSED_UNIT (not timed against the two-launch form on this path)
            self.call("cosmos_elbo_grads", a)
            handle = allreduce(self.gsum) if allreduce is not None else None
            if handle is not None and hasattr(handle, "wait") and a.fuse_adam:
                self._pending = (a, handle)
            else:
                if handle is not None and hasattr(handle, "wait"):
                    handle.wait()
                self._tail_reduced(a, None)
        self.adam_step += 1

    def _tail_reduced(self, a, next_args):
        """Everything of step `a` after the all-reduce (one launch), plus the global draws of `next_args`."""
        nxt = None if next_args is None else C.byref(next_args)
        _lib.check(self.lib.tq_cosmos_tail_reduced(C.byref(a), nxt, self._stream()), "tq_cosmos_tail_reduced")

    def _finish_pending(self, next_args=None):
        """Global tail of a step whose all-reduce was left in flight."""
        if self._pending is None:
            return
        a, handle = self._pending
        self._pending = None
        handle.wait()  # the current stream waits for the collective
        self._tail_reduced(a, next_args)

    def _finish_tail(self):
        """Tail of a pipelined full-batch step (tq_cosmos_step_overlapped left it pending)."""
        if self._tail_args is not None:
            a, self._tail_args = self._tail_args, None
            self.call("cosmos_tail", a)

    def join(self):
        """Complete deferred work: the pending tail of a pipelined step, or the global tail of a sharded step that waits
        for its all-reduce.  Every read-out of elbo_out / parameters goes through here."""
        self._finish_pending()
        self._finish_tail()
        self._catch_up_all()

    # -- named views -----------------------------------------------------------------------------------
    def named(self, which="params"):
        self.join()  # finish a pending global tail first: the views then hold the parameters after the last step
        return self.layout.views(getattr(self, which))
