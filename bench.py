"""
Benchmark of the cosmos SVI hot path on MI355X.

    python bench.py                                  # 1 GPU, BASELINE config c2, K steps x 5 blocks
    python bench.py --gpus N --steps K --warmup W    # starts its own N rank processes (one per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (same ranks, outer launcher)

Workload (BASELINE.json configs[1] = "c2"): cosmos K=2, 400 AOIs x 1000 frames, P=14, synthetic data with the law of
tapqir/utils/simulate.py and the reference test-suite parameters; one *step* = one complete SVI update (guide draws ->
ELBO -> gradients -> dense Adam) over the full batch held by the rank.  With N ranks every rank holds its own shard
(AOI-sharded, weak scaling: `--config c2` = 400 x 1000 per GPU) and the only communication is ONE all-reduce of the
cross-unit sums per step.  `--config c3` / `c5` / `c4` put the per-GPU shard of the other BASELINE configs on every rank
(c3: 400 x 4000, = BASELINE c3 itself at N = 8; c5: K=3 P=20 250 x 2000, = c5 itself at N = 4; c4: crosstalk
400 x 1000 x 2 channels); with `--config auto` (default) the headline is c2 per GPU and a run at N = 8 / N = 4 appends the
c3 / c5 figures as `north_star_configs`.

Prints ONE JSON line (rank 0).  `value` = AOI-frames/s over all ranks (median of `--blocks` timed blocks of exactly
`--steps` steps, each bracketed by barrier + synchronize, max over ranks); also reported: ELBO steps/s, the reference's
default-minibatch operating point, the trained-parameter regime, the HBM roofline of the fused spot-render + log-prob
kernel (HIP-event timing on its own stream) and the CPU baseline (oracle = dense-torch restatement, timed on this box's
host cores, float64 as `tapqir fit` and float32 for information).
"""

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X nominal HBM3E bandwidth (MI355X_MICROARCH.md); ~6300 GB/s achievable

# per-GPU shards of the BASELINE.json configs: (model, K, P, AOIs per GPU, frames, channels, ranks at which it IS the config)
CONFIGS = {
    "c1": ("cosmos", 1, 14, 50, 100, 1, 1),  # the reference's own CPU-runnable case (plumbing; 5 000 units: launch-bound on a GPU)
    "c2": ("cosmos", 2, 14, 400, 1000, 1, 1),
    "c3": ("cosmos", 2, 14, 400, 4000, 1, 8),
    "c4": ("crosstalk", 2, 14, 400, 1000, 2, 1),
    "c5": ("cosmos", 3, 20, 250, 2000, 1, 4),
}


def fwd_bytes_per_unit(K, P):
    """tile + xy + (h,w,x,y per spot; b) + 2^K outputs, fp32."""
    return 4 * P * P + 8 + 4 * (4 * K + 1) + 4 * 2**K


def bwd_bytes_per_unit(K, P):
    """The kernel a step runs also reads the K m_probs logits and writes the 2+4K pathwise gradients."""
    return fwd_bytes_per_unit(K, P) + 4 * K + 4 * (2 + 4 * K)


def step_bytes_per_unit(K, P):
    """fused step with dense Adam, gradients never round-tripping."""
    return 4 * P * P + 8 + 6 * 4 * (8 * K + 2)


PROFILE_ROUNDS = ("r03", "r02", "r01")


def _profile_json(suffix):
    """Newest committed profiles/rNN_<suffix> (written by scripts/gpu_r03_profiles.sh from rocprofv3 runs of THIS script)."""
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", f"{rnd}_{suffix}")
        try:
            return json.load(open(path)), f"profiles/{rnd}_{suffix}"
        except OSError:
            continue
    return None, None


def pmc_traffic(K, P, units, backward):
    """HBM bytes per launch of the stand-alone log-prob kernel from the committed PMC passes (rocprofv3 FETCH_SIZE /
    WRITE_SIZE in separate runs, gfx950 correction applied; profiles/*_pmc_traffic.json says how).  PMC collection cannot
    run inside this process, so the figure is the measured one for this exact kernel and shape, else None."""
    d, name = _profile_json("pmc_traffic.json")
    if d is None or (d.get("K"), d.get("P"), d.get("units")) != (K, P, units):
        return None, None
    for kname, k in d["kernels"].items():
        if kname.endswith(f"<{K}, {P}, {'true' if backward else 'false'}>"):
            return k["traffic_bytes"], name
    return None, None


def step_traffic(cfg, units):
    """Per-launch HBM bytes of every kernel of one STEP of config `cfg`, and their sum, from the committed PMC passes over
    `bench.py --quick` (profiles/*_pmc_step_traffic.json, scripts/make_step_traffic.py).  None if not collected for this
    config and size."""
    d, name = _profile_json("pmc_step_traffic.json")
    if d is None:
        return None, None
    c = d["configs"].get(cfg)
    if not c or c.get("units") != units:
        return None, None
    return c, name


def trace_duration(cfg, kernel_substr):
    """Average duration (ms) of a kernel in the committed rocprofv3 kernel trace of `bench.py` for config `cfg`
    (profiles/*_kernel_trace.json, scripts/gpu_r03_profiles.sh), else None."""
    d, name = _profile_json("kernel_trace.json")
    if d is None:
        return None, None
    for k, v in d.get(cfg, {}).items():
        if kernel_substr in k:
            return v["avg_us"] * 1e-3, name
    return None, None


def pmc_valu_issue(kernel_tag, units, avg_launch_s):
    """VALU issue figures of the dominant kernel from the committed PMC passes (profiles/*_pmc_valu.json; separate
    rocprofv3 --pmc runs, scripts/gpu_step_pmc.sh): the kernels whose HBM fraction is low are bound by instruction issue,
    and this is the roof they are at.  `busy_at_nominal_clock` prices the measured launch of THIS run at 2.4 GHz."""
    prof = os.path.join(ROOT, "profiles")
    for name in ("r03_pmc_valu.json", "r02_pmc_valu.json"):
        try:
            d = json.load(open(os.path.join(prof, name)))
        except OSError:
            continue
        c = d["configs"].get(kernel_tag)
        if c and c["units"] == units:
            cycles_per_simd = c["SQ_ACTIVE_INST_VALU"] * 4 / 1024
            return {"bound": "valu_issue", "source": f"profiles/{name}", "kernel": c["kernel"],
                    "valu_instructions_per_launch": c["SQ_INSTS_VALU"], "of_which_transcendental": c["SQ_INSTS_VALU_TRANS_F32"],
                    "issue_cycles_per_simd": cycles_per_simd, "busy_in_profiled_launch": c["valu_busy"],
                    "busy_at_nominal_clock": cycles_per_simd / (avg_launch_s * 2.4e9),
                    "note": "fraction of the launch during which a SIMD issues a VALU instruction (SQ_ACTIVE_INST_VALU x 4 / 1024 "
                            "SIMDs over the launch's cycles); 1.0 is the roof of an issue-bound kernel"}
    return None


# ---------------------------------------------------------------------------------------------------------------------
# parent: start one process per GPU.  Nothing in this function (or before it in main) touches the GPU.
# ---------------------------------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    import torch

    n = args.gpus
    have = torch.cuda.device_count()  # does not initialise the HIP runtime
    if have < n:
        print(f"bench.py: --gpus {n} but this node exposes {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = subprocess.PIPE if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out))
    line, _ = procs[0].communicate()
    rc = procs[0].returncode
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            p.wait(max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()  # exactly the process this launcher started
        rc = rc or p.returncode
    sys.stdout.write(line.decode())
    sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------------------------------------------
class Problem:
    """Synthetic data + engine of one rank for one config."""

    def __init__(self, cfg, rank, world, dev, offsets="sim", aois=None, frames=None):
        import torch

        from tapqir_amd.models.cosmos import initial_values
        from tapqir_amd.models.crosstalk import crosstalk_initial_values
        from tapqir_amd.models.engine import CosmosEngine
        from tapqir_amd.utils.dataset import CosmosDataset
        from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

        model, K, P, N, F, Cc, at = CONFIGS[cfg]
        N, F = aois or N, frames or F
        self.cfg, self.model, self.K, self.P, self.N, self.F, self.C, self.is_config_at = cfg, model, K, P, N, F, Cc, at
        self.world, self.rank = world, rank
        xt = model == "crosstalk"

        class _M:  # minimal "model" for simulate(): K and device
            pass

        _M.K, _M.device = K, dev
        sim_params = dict(TEST_PARAMS, alpha=[[0.9, 0.1], [0.2, 0.8]]) if xt else TEST_PARAMS
        data = simulate(_M, N, F, Cc, P, seed=1000 + rank, params=sim_params)
        if offsets == "hist":
            s = torch.arange(70.0, 120.0)
            w = torch.minimum(s - 69.0, 120.0 - s)
            data = CosmosDataset(data.images, data.xy, data.is_ontarget, offset_samples=s, offset_weights=w / w.sum())
        self.data = data
        self.eng = CosmosEngine(data, K=K, device=dev, seed=7, n_offset=rank * N, Nt_global=world * N, crosstalk=xt)
        # initial parameter values of the reference (cosmos.py:471-598, crosstalk.py:424-455)
        self.eng.layout.set_constrained(self.eng.params, (crosstalk_initial_values if xt else initial_values)(self.eng, data))
        self.offsets = offsets

    def workload(self):
        at = self.is_config_at
        same = " (= BASELINE config %s itself)" % self.cfg if self.world == at else \
            f" (per-GPU shard of BASELINE config {self.cfg}, which spans {at} GPUs)" if at > 1 else ""
        return (f"{self.model} K={self.K}, {self.N} AOIs x {self.F} frames x {self.C} channel(s) per GPU, P={self.P}, "
                f"full-batch SVI step (sample guide, ELBO, gradients, dense Adam); offsets={self.offsets} "
                f"(O={self.eng.O} after merging){same}")


class Runner:
    def __init__(self, use_dist, dev):
        self.use_dist, self.dev = use_dist, dev
        self.allreduce = None
        self.backend = "none"
        if use_dist:
            import torch.distributed as dist

            direct = None
            if os.environ.get("TAPQIR_AMD_RCCL_DIRECT", "1") != "0":
                # ncclAllReduce of RCCL on the launch stream itself (tapqir_amd/rccl.py): stream order is the only dependency.
                # Checked once against the process group before it is trusted with the timed steps (no N > 1 hardware has run
                # this path yet); any failure on any rank -> torch's path on all ranks.
                from tapqir_amd.rccl import RcclDirect

                sys.stdout.flush()
                saved = os.dup(1)  # (RCCL may print its banner on stdout when a communicator is created: keep it off the JSON line)
                os.dup2(2, 1)
                try:
                    direct = RcclDirect.checked(device=dev)
                finally:
                    sys.stdout.flush()
                    os.dup2(saved, 1)
                    os.close(saved)
            if direct is not None:
                self.allreduce = direct
                self.backend = direct.backend
            else:
                # torch's process group (its own stream), left in flight: the engine overlaps it with the next step's local
                # guide sampling (full-batch steps)
                self.allreduce = lambda t: dist.all_reduce(t, async_op=True)
                self.backend = dist.get_backend()

    def run(self, eng, n, ndx=None, fdx=None):
        for _ in range(n):
            eng.step(ndx, fdx, allreduce=self.allreduce)
        eng.join()  # the last step's deferred global tail belongs to the timed region

    def barrier(self):
        import torch

        if self.use_dist:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    def timed_blocks(self, eng, steps, warmup, blocks):
        """`blocks` timed regions of exactly `steps` steps, each bracketed by barrier + synchronize; per block the MAX
        over ranks.  Returns (median block seconds, all block seconds, this run's per-rank seconds of the median block)."""
        import torch

        self.run(eng, warmup)
        per_block, per_rank = [], []
        for _ in range(blocks):
            self.barrier()
            t0 = time.perf_counter()
            self.run(eng, steps)
            self.barrier()
            dt = time.perf_counter() - t0
            ranks = [dt]
            if self.use_dist:
                import torch.distributed as dist

                tt = torch.tensor([dt], device=self.dev, dtype=torch.float64)
                allt = [torch.zeros_like(tt) for _ in range(dist.get_world_size())]
                dist.all_gather(allt, tt)
                ranks = [float(t) for t in allt]
            per_block.append(max(ranks))
            per_rank.append(ranks)
        order = sorted(range(blocks), key=lambda i: per_block[i])
        mid = order[(blocks - 1) // 2]
        return per_block[mid], per_block, per_rank[mid]


def measured_copy_bandwidth(dev, nbytes=1 << 30, reps=10):
    """Device-to-device copy rate (read + write bytes per second, GB/s): what this box's HBM delivers to a plain
    streaming kernel; quoted next to the nominal 8 TB/s peak."""
    import torch

    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    e1.synchronize()
    return 2 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def time_pixel_kernel(eng, launches, backward):
    """Average duration of the fused render+log-prob kernel (cosmos or crosstalk), HIP events on its launch stream."""
    import torch

    from tapqir_amd import _lib

    a = eng.make_args()
    # fresh guide draws in the (possibly re-allocated) full-batch workspace
    eng.call("cosmos_sample_globals", a)
    eng.call("cosmos_sample_locals", a)
    K, M = eng.K, 1 << eng.K
    B = eng.Nt * eng.F * eng.C
    xt = eng.crosstalk
    if not xt:
        k = eng.ksmogn_args(backward)  # the form (pixel_mode) the engine chose for this box
    else:
        k = _lib.XtalkArgs()
        p = _lib.ptr
        k.images, k.images_il, k.xy, k.ndx, k.fdx = p(eng.images), p(eng.images_il), p(eng.xy), None, None
        k.nb_full, k.il_min_units = eng.Nt, eng.il_min_units
        k.pixstats = p(eng.pixstats)
        lat = eng.lat
        f = lambda row: lat.data_ptr() + 4 * row * B
        k.background, k.height, k.width, k.x, k.y = f(0), f(1), f(1 + K), f(1 + 2 * K), f(1 + 3 * K)
        k.gain = eng.globals.data_ptr()
        k.offset_samples, k.offset_logits = p(eng.offset_samples), p(eng.offset_logits)
        k.gout, k.m_logit, k.aoi_mask = None, p(eng.params), p(eng._mask_arg)
        pix = eng.pix
        g = lambda row: pix.data_ptr() + 4 * row * B
        k.ll = g(0)
        k.alpha = eng.globals.data_ptr() + 4 * 21  # TqGlobals.alpha (tq_site.h)
        k.ll_joint = None
        k.ell_excess = g(M + 2 + 4 * K)
        k.g_alpha = g(M + 3 + 4 * K)
        if backward:
            k.g_background, k.g_gain = g(M), g(M + 1)
            k.g_height, k.g_width, k.g_x, k.g_y = g(M + 2), g(M + 2 + K), g(M + 2 + 2 * K), g(M + 2 + 3 * K)
        k.m_kstride = B
        k.nb, k.fb, k.C, k.F, k.P, k.K, k.O = eng.Nt, eng.F, eng.C, eng.F, eng.P, K, eng.O
        k.scale = 1.0
    fn = eng.lib.tq_ksmogn_crosstalk_log_prob if xt else eng.lib.tq_ksmogn_log_prob
    stream = torch.cuda.current_stream()
    sp = C.c_void_p(stream.cuda_stream)
    for _ in range(3):
        _lib.check(fn(C.byref(k), sp), "pixel kernel")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(launches):
        fn(C.byref(k), sp)
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / launches * 1e-3  # seconds


def cpu_baseline(data, K, nb, fb, steps=5, warmup=2, dtype="float64", threads=None):
    """Oracle (dense torch = the tensor program Pyro would run) on the host cores."""
    import torch

    from oracle import cosmos as oc

    with oc.working_dtype(getattr(torch, dtype)):
        od = oc.OracleData(data.images[:nb, :fb].cpu(), data.xy[:nb, :fb].cpu(), data.is_ontarget[:nb].cpu(),
                           data.offset.samples.cpu(), data.offset.weights.cpu())
        o = oc.CosmosOracle(od, K=K, eps=float(torch.finfo(torch.float32).eps) if dtype == "float32" else None)
        o.init_parameters()
        o.make_optim(lr=0.005)
        nd, fd = torch.arange(nb), torch.arange(fb)
        # torch's intra-op pool does not scale to hundreds of host threads on these tensor sizes:
        # take the thread count that runs this step fastest (one probe step each) and say which it was
        if threads is None:
            best = None
            for nt in sorted({t for t in (8, 16, 32, 64, os.cpu_count()) if t <= os.cpu_count()}):
                torch.set_num_threads(nt)
                o.step(nd, fd)
                t0 = time.perf_counter()
                o.step(nd, fd)
                dt = time.perf_counter() - t0
                if best is None or dt < best[0]:
                    best = (dt, nt)
            threads = best[1]
        torch.set_num_threads(threads)
        ts = []
        for it in range(warmup + steps):
            t0 = time.perf_counter()
            loss = o.step(nd, fd)
            if it >= warmup:
                ts.append(time.perf_counter() - t0)
        assert loss == loss, "CPU baseline produced a NaN loss"
    ts.sort()
    med = ts[len(ts) // 2]
    return {"value": nb * fb / med, "unit": "AOI-frames/s", "cores": torch.get_num_threads(),
            "host_cpus": os.cpu_count(), "kind": "port", "dtype": dtype,
            "sample": f"oracle dense-torch {dtype} full SVI step, nb={nb} x fb={fb} units of the same data, "
                      f"median of {steps} steps after {warmup} warm-up ({med:.2f} s/step = {1 / med:.3f} steps/s)",
            "steps_per_sec_at_sample": 1 / med}


def time_fused_kernel(eng, launches):
    """Average duration of tq_pixel_unit_kernel (pixel kernel + per-unit terms + Adam of a full-batch step in one launch),
    HIP events on its launch stream.  Every launch applies an Adam step to the local parameters: they are put back."""
    import torch

    from tapqir_amd import _lib

    eng.join()
    saved = (eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone())
    a = eng.make_args()
    a.fuse_adam, a.pixel_mode, a.last_step = 1, 2, None
    eng.call("cosmos_sample_globals", a)
    eng.call("cosmos_sample_locals", a)
    for _ in range(2):
        eng.call("cosmos_pixel_unit", a)
    # As in a step, the local sampling launch runs between two launches of this kernel (back to back with itself the
    # kernel also waits for its predecessor's 86 MB of parameter writes to drain).  Timed as the difference between
    # `launches` x (sampling, this kernel) and `launches` x sampling, each as one uninterrupted sequence: an event
    # between two launches would add the dispatch latency that queued launches hide.
    def sequence(with_kernel):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            eng.call("cosmos_sample_locals", a)
            if with_kernel:
                eng.call("cosmos_pixel_unit", a)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1)

    sequence(True)
    total_ms = sequence(True) - sequence(False)
    eng.params.copy_(saved[0])
    eng.exp_avg.copy_(saved[1])
    eng.exp_avg_sq.copy_(saved[2])
    return total_ms / launches * 1e-3


def roofline_block(pb, ms_per_step, dev):
    """`roofline` of the bench line: the DOMINANT kernel of the timed step -- the fused pixel + per-unit launch where the
    step runs it (c1 / c2 / c3), else the log-prob (pixel) kernel -- at its algorithmic bytes, with its duration measured here
    (HIP events on the launch stream) next to the one in the committed rocprofv3 trace of this script, its PMC traffic, the
    traffic of the whole step and the stand-alone log-prob kernel as sub-objects."""
    import torch

    eng, K, P = pb.eng, pb.K, pb.P
    units = pb.N * pb.F * (1 if eng.crosstalk else pb.C)  # crosstalk: one (C, P, P) tile per AOI-frame
    t_fb = time_pixel_kernel(eng, 20, backward=True)
    t_f = time_pixel_kernel(eng, 20, backward=False)
    if eng.crosstalk:
        # one launch covers all C channels of an AOI-frame: C tiles + C (xy, b) + Q dyes x K spots + the 2^K marginals per dye
        bpu = pb.C * (4 * P * P + 8 + 4) + pb.C * (16 * K + 4 * 2**K)
        kernel = "tq_xtalk_il_kernel<K,P,bwd> (coupled-dye render + log-prob of the 2^(KQ) joint combinations + pathwise grads)"
    else:
        bpu = fwd_bytes_per_unit(K, P)
        kernel = (("tq_ksmogn_il2p_kernel<K,P,bwd> (persistent form" if eng.pixel_mode else "tq_ksmogn_il2_kernel<K,P,bwd> (one wave per tile")
                  + "; fused render + log-prob + pathwise grads, packed lane-per-unit; form chosen by timing both on this box: "
                  + str([round(t, 4) for t in getattr(eng, "pixel_times_ms", [])]) + " ms)") \
            if eng.O == 1 else "tq_ksmogn_il2m_kernel<K,bwd> (offset-histogram form of the same kernel)"
    ach = units * bpu / t_fb / 1e9
    traffic, tsrc = (pmc_traffic(K, P, units, True) if (pb.offsets == "sim" and not eng.crosstalk) else (None, None))
    sb = step_bytes_per_unit(K, P)
    tot_units = pb.N * pb.F * pb.C
    logprob = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
               "traffic": traffic,
               "traffic_unit": f"bytes/launch (PMC, {tsrc}); algorithmic = bytes_per_unit x units_per_launch",
               "kernel": kernel, "bytes_per_unit": bpu, "units_per_launch": units, "avg_launch_ms": t_fb * 1e3,
               "timing": "stand-alone launches of the kernel, HIP events on the launch stream",
               "forward_only": {"avg_launch_ms": t_f * 1e3, "achieved": units * bpu / t_f / 1e9,
                                "frac": units * bpu / t_f / 1e9 / HBM_PEAK_GBS}}
    if not eng.crosstalk:
        bb = bwd_bytes_per_unit(K, P)
        logprob["with_gradient_outputs"] = {"bytes_per_unit": bb, "achieved": units * bb / t_fb / 1e9,
                                            "frac": units * bb / t_fb / 1e9 / HBM_PEAK_GBS,
                                            "note": "bytes_per_unit + K m_probs logits read + (2+4K) gradient rows written"}
    cfg_tag = pb.cfg if (pb.N, pb.F) == CONFIGS[pb.cfg][3:5] and pb.offsets == "sim" else None
    st, stsrc = step_traffic(cfg_tag, tot_units) if cfg_tag else (None, None)
    fused = bool(getattr(eng, "fuse_unit", False)) and eng._fusable()
    if fused:
        # what the timed steps launch: pixel kernel + per-unit terms + Adam as one kernel, whose algorithmic bytes are those
        # of the whole step (tile + target position + read/write of every local parameter and moment)
        t_pu = time_fused_kernel(eng, 10)
        tr, trsrc = trace_duration(cfg_tag, "tq_pixel_unit_kernel") if cfg_tag else (None, None)
        out = {"bound": "hbm", "achieved": tot_units * sb / t_pu / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": tot_units * sb / t_pu / 1e9 / HBM_PEAK_GBS,
               "kernel": "tq_pixel_unit_kernel<K,P>: the dominant launch of the timed step (render + log-prob + pathwise grads of a "
                         "tile of 64 units, then the per-unit ELBO terms, gradients and Adam of the same units in the same wave; "
                         + (f"chosen over two launches by timing both on this box: {[round(t, 4) for t in eng.step_times_ms]} ms per step)"
                            if getattr(eng, "step_times_ms", None) else "the default wherever the step qualifies; TAPQIR_AMD_FUSE_UNIT=auto times both forms)"),
               "bytes_per_unit": sb, "units_per_launch": tot_units, "avg_launch_ms": t_pu * 1e3,
               "timing": "HIP events on the launch stream: (launches x [sampling, this kernel]) - (launches x sampling), each an "
                         "uninterrupted sequence, so that the kernel is timed between the launches it runs between in a step",
               "trace": {"avg_launch_ms": tr, "source": trsrc,
                         "note": "kernel-only duration of the same kernel in the committed rocprofv3 --kernel-trace of this script"},
               "traffic": (st["kernels"].get("tq_pixel_unit_kernel", {}).get("traffic_bytes") if st else None),
               "traffic_unit": f"bytes/launch of this kernel (PMC FETCH_SIZE x 2 KiB + WRITE_SIZE KiB in separate passes over this script, {stsrc}); "
                               "algorithmic = bytes_per_unit x units_per_launch"}
    else:
        out = dict(logprob)
        out["kernel"] = "dominant launch of the timed step: " + kernel
        if st:
            k0 = next((v for k, v in st["kernels"].items() if "ksmogn" in k or "xtalk" in k), None)
            if k0 and traffic is None:
                out["traffic"], out["traffic_unit"] = k0["traffic_bytes"], f"bytes/launch (PMC passes over this script, {stsrc})"
    out["logprob_kernel"] = logprob
    out["whole_step"] = {"bytes_per_unit": sb, "achieved": tot_units * sb / (ms_per_step * 1e-3) / 1e9,
                         "frac": tot_units * sb / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "note": "algorithmic bytes of the step (tile, target position, every local parameter and Adam moment read and "
                                 "written once) over the driver-timed step"}
    if st:
        out["step_traffic"] = {"bytes": st["step_traffic_bytes"], "algorithmic": tot_units * sb,
                               "ratio": st["step_traffic_bytes"] / (tot_units * sb), "source": stsrc,
                               "per_kernel": {k: v["traffic_bytes"] for k, v in st["kernels"].items()},
                               "note": "HBM bytes of ALL launches of one step (PMC) against the step's algorithmic bytes: the excess is the "
                                       "hand-off of the guide draws and site terms between the two launches"}
    tag = "c4" if eng.crosstalk else ("hist" if eng.O > 1 else ("c5" if (K, P) == (3, 20) else ("c2" if (K, P) == (2, 14) else None)))
    vi = pmc_valu_issue(tag, units, t_fb) if tag else None
    if vi:
        out["valu_issue"] = vi
    if eng.O > 1:
        # the offset-histogram kernel is bound by the transcendental pipe, not by HBM: (K+1) + 1 exp2/log2 per
        # (offset, pixel); peak = v_exp_f32 issue rate of profiles/r01_valu_issue_rates.txt (8 cycles per wave64) x 1024 SIMDs
        per_unit = P * P * eng.O * (K + 2)
        peak = 1024 * 64 / 8 * 2.4e9 / 1e12
        out["transcendental"] = {"bound": "transcendental", "per_unit": per_unit,
                                 "achieved": units * per_unit / t_fb / 1e12, "peak": peak, "unit": "Ttrans/s",
                                 "frac": units * per_unit / t_fb / 1e12 / peak}
    bw = measured_copy_bandwidth(dev)
    out["peak_measured_copy"] = bw
    out["frac_of_measured_copy"] = out["achieved"] / bw
    logprob["forward_only"]["frac_of_measured_copy"] = logprob["forward_only"]["achieved"] / bw
    return out


def headline(pb, runner, args, prewarm=True):
    """Timed full-batch steps of one problem -> dict of the headline fields."""
    eng = pb.eng
    if prewarm:
        # The first ~1000 kernel launches of a process end with ONE host-side stall of ~80 ms inside the HIP runtime
        # (scripts/diag_hiccup.py; the GPU is idle meanwhile).  Push the process past that point first.
        runner.run(eng, 400)
        runner.barrier()
    med, blocks, per_rank = runner.timed_blocks(eng, args.steps, args.warmup, args.blocks)
    ms = med / args.steps * 1e3
    units = pb.world * pb.N * pb.F
    import torch

    eng.join()
    assert torch.isfinite(eng.params).all(), "non-finite parameters after the timed steps"
    return {"value": units / (med / args.steps), "ms_per_step": ms, "elbo_steps_per_sec": 1e3 / ms,
            "block_ms_per_step": [b / args.steps * 1e3 for b in blocks],
            "per_rank_ms_per_step": [r / args.steps * 1e3 for r in per_rank],
            "final_elbo": float(eng.elbo_out[0]), "workload": pb.workload()}


def worker(args):
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    use_dist = world > 1 or args.force_dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on STDOUT when the communicator is created; stdout carries the ONE JSON line of
        # this script, so file descriptor 1 points at stderr until the first collective has run
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)
            t = torch.zeros(1, device=dev)
            dist.all_reduce(t)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
        world = dist.get_world_size()  # what the process group says, not the environment
        rank = dist.get_rank()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s)")

    runner = Runner(use_dist, dev)
    cfg = "c2" if args.config == "auto" else args.config
    pb = Problem(cfg, rank, world, dev, offsets=args.offsets, aois=args.aois, frames=args.frames)
    eng = pb.eng
    if args.pre_steps > 0:
        # untimed steps of fitting before anything is measured (profiling runs of the trained-parameter regime)
        runner.run(eng, args.pre_steps)
        runner.barrier()
    h = headline(pb, runner, args)
    # host side of a step (outside the timed region): time to ENQUEUE 20 steps on an idle queue; if it approaches
    # ms_per_step the launch path, not the GPU, sets the pace
    runner.barrier()
    th = time.perf_counter()
    for _ in range(20):
        eng.step(allreduce=runner.allreduce)
    host_ms = (time.perf_counter() - th) / 20 * 1e3
    eng.join()
    runner.barrier()

    out = {
        "metric": f"{pb.model} SVI AOI-frames/s (= ELBO steps/s x nb x fb), K={pb.K} P={pb.P} full batch",
        "value": h["value"], "unit": "AOI-frames/s", "n_gpus": world,
        "rccl_ranks": (dist.get_world_size() if use_dist else 1), "backend": runner.backend,
        "steps": args.steps, "warmup": args.warmup, "blocks": args.blocks,
        "ms_per_step": h["ms_per_step"], "block_ms_per_step": h["block_ms_per_step"],
        "per_rank_ms_per_step": h["per_rank_ms_per_step"], "host_enqueue_ms_per_step": host_ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": h["workload"], "baseline_config": cfg, "nb": pb.N, "fb": pb.F,
                   "aoi_sharding": f"{world} x {pb.N} AOIs", "pre_steps": args.pre_steps},
        "elbo_steps_per_sec": h["elbo_steps_per_sec"], "final_elbo": h["final_elbo"],
    }

    if world == 1 and args.profile_minibatch > 0:
        # profiling runs: N steps at the reference's default minibatch (10 x 512, fresh subsample every step), untimed
        nb, fb = min(10, pb.N), min(512, pb.F)
        g = torch.Generator(device="cpu").manual_seed(0)
        for _ in range(args.profile_minibatch):
            eng.step(torch.randperm(pb.N, generator=g)[:nb], torch.randperm(pb.F, generator=g)[:fb])
        eng.join()
        torch.cuda.synchronize()

    if world == 1 and not args.quick:
        # ---- roofline of the dominant kernel (same parameter regime as the headline steps) -----------
        out["roofline"] = roofline_block(pb, h["ms_per_step"], dev)
        # ---- reference default minibatch operating point (main.py:1429-1430): nb=10, fb=512 ----------
        nb, fb = min(10, pb.N), min(512, pb.F)
        g = torch.Generator(device="cpu").manual_seed(0)
        idx = [(torch.randperm(pb.N, generator=g)[:nb], torch.randperm(pb.F, generator=g)[:fb]) for _ in range(100)]
        for nd, fd in idx[:10]:
            eng.step(nd, fd)
        mbs = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for nd, fd in idx:
                eng.step(nd, fd)
            eng.join()
            torch.cuda.synchronize()
            mbs.append((time.perf_counter() - t0) / len(idx))
        mb = sorted(mbs)[2]
        out["minibatch_10x512"] = {"ms_per_step": mb * 1e3, "steps_per_sec": 1 / mb, "aoi_frames_per_sec": nb * fb / mb,
                                   "protocol": "median of 5 blocks of 100 steps, fresh random subsample every step"}
        # the path of Model.run: the subsample of step t + 1 drawn on the device by the launch of step t
        # (CosmosEngine.step_subsampled; falls back to the figures above where that path does not apply)
        if eng.step_subsampled(nb, fb, g):
            for _ in range(20):
                eng.step_subsampled(nb, fb, g)
            dss = []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100):
                    eng.step_subsampled(nb, fb, g)
                eng.join()
                torch.cuda.synchronize()
                dss.append((time.perf_counter() - t0) / 100)
            ds = sorted(dss)[2]
            out["minibatch_10x512"]["device_subsampled"] = {"ms_per_step": ds * 1e3, "steps_per_sec": 1 / ds,
                                                            "aoi_frames_per_sec": nb * fb / ds}
        # ---- trained-parameter regime: guide concentrations shrink as the fit converges and other regimes of the
        # implicit reparameterisation gradients take over ------------------------------------
        if args.trained_steps > 0 and cfg == "c2" and args.offsets == "sim":
            runner.run(eng, args.trained_steps)
            med, blocks, _ = runner.timed_blocks(eng, args.steps, 0, args.blocks)
            out["trained_regime"] = {"after_steps": args.trained_steps, "ms_per_step": med / args.steps * 1e3,
                                     "value": pb.N * pb.F / (med / args.steps), "final_elbo": float(eng.elbo_out[0])}
        if not args.no_cpu and pb.model == "cosmos":
            out["cpu_baseline"] = cpu_baseline(pb.data, pb.K, min(10, pb.N), min(512, pb.F))
            f32 = cpu_baseline(pb.data, pb.K, min(10, pb.N), min(512, pb.F), dtype="float32",
                               threads=out["cpu_baseline"]["cores"])
            out["cpu_baseline"]["float32"] = {k: f32[k] for k in ("value", "steps_per_sec_at_sample", "sample", "cores")}

    # ---- the other BASELINE configs at the rank counts they are quoted on (driver's SCALE run: N = 8 -> c3, N = 4 -> c5)
    extra = []
    if args.config == "auto" and args.offsets == "sim" and not args.quick:
        extra = {8: ["c3"], 4: ["c5"]}.get(world, [])
    if args.also:
        extra += [c for c in args.also.split(",") if c and c != cfg]
    if extra:
        out["north_star_configs"] = {}
        del pb, eng
        torch.cuda.empty_cache()
        for c in extra:
            pe = Problem(c, rank, world, dev)
            he = headline(pe, runner, args, prewarm=False)
            entry = {k: he[k] for k in ("value", "ms_per_step", "block_ms_per_step", "per_rank_ms_per_step", "workload")}
            entry.update(unit="AOI-frames/s", n_gpus=world, is_the_config=(world == pe.is_config_at))
            if world == 1:
                entry["roofline"] = roofline_block(pe, he["ms_per_step"], dev)
            out["north_star_configs"][c] = entry
            del pe
            torch.cuda.empty_cache()

    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of --steps steps; the median block is reported")
    ap.add_argument("--config", default="auto", choices=["auto"] + sorted(CONFIGS),
                    help="BASELINE.json config whose per-GPU shard every rank holds (auto = c2, plus c3 at 8 ranks / c5 at 4)")
    ap.add_argument("--also", default="", help="comma-separated configs to time after the headline (north_star_configs)")
    ap.add_argument("--aois", type=int, default=None, help="override the AOIs per GPU of the config")
    ap.add_argument("--frames", type=int, default=None, help="override the frames of the config")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--quick", action="store_true", help="headline only (no minibatch / roofline / CPU legs)")
    ap.add_argument("--trained-steps", type=int, default=4000, help="steps before the trained-regime leg (0 = skip)")
    ap.add_argument("--profile-minibatch", type=int, default=0, help="untimed default-minibatch steps after the headline (profiling runs)")
    ap.add_argument("--pre-steps", type=int, default=0, help="untimed full-batch steps before the headline (profiling the trained regime)")
    ap.add_argument("--offsets", default="sim", choices=["sim", "hist"])
    ap.add_argument("--model", default=None, choices=["cosmos", "crosstalk"], help="crosstalk = --config c4")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: take the multi-GPU code path (process group, staged step, all-reduce) with one rank")
    args = ap.parse_args()
    if args.model == "crosstalk":
        args.config = "c4"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, sys.argv[1:])  # parent: never touches the GPU
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
