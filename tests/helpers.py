"""Shared test utilities: problem construction, oracle <-> engine plumbing, host-check loader."""

import ctypes as C
import os
import subprocess

import torch

from oracle.cosmos import CosmosOracle, OracleData
from oracle.crosstalk import CrosstalkOracle
from tapqir_amd import _lib
from tapqir_amd.models.engine import CosmosEngine as HipEngine
from tapqir_amd.utils.dataset import CosmosDataset
from tapqir_amd.utils.simulate import TEST_PARAMS, simulate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HC_DIR = os.path.join(ROOT, "tests", "hostcheck")
EPS32 = float(torch.finfo(torch.float32).eps)

_hc = None

# stages of a step whose latent draws are supplied by the test (make_args(draw_globals=False)):
# tables from the given global base draws, site terms of the given local draws, ELBO + gradients
GIVEN_STAGES = ("cosmos_sample_globals", "cosmos_sample_locals", "cosmos_elbo_grads", "cosmos_globals_grad")


def load_hostcheck():
    """g++ build of the kernels' inline math, driven on host memory (tests only)."""
    global _hc
    if _hc is not None:
        return _hc
    so = os.path.join(HC_DIR, "libtq_hostcheck.so")
    src = os.path.join(HC_DIR, "hostcheck.cpp")
    hdrs = [os.path.join(ROOT, "tapqir_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "tapqir_amd", "csrc"))
            if f.endswith(".h")] + [os.path.join(ROOT, "include", "tapqir_hip.h"), src]
    if not os.path.exists(so) or any(os.path.getmtime(h) > os.path.getmtime(so) for h in hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                               "-o", so, src])
    lib = C.CDLL(so)
    lib.hc_globals_size.restype = C.c_int64
    lib.hc_gbase_size.restype = C.c_int64
    lib.hc_ksmogn_log_prob.argtypes = [C.POINTER(_lib.KsmognArgs)]
    lib.hc_ksmogn_crosstalk_log_prob.argtypes = [C.POINTER(_lib.XtalkArgs)]
    lib.hc_image_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
    lib.hc_image_stats.restype = None
    lib.hc_snr_chi2.argtypes = [C.POINTER(_lib.SnrArgs)]
    lib.hc_snr_chi2.restype = None
    lib.hc_ksmogn_rsample.argtypes = [C.POINTER(_lib.RsampleArgs)]
    lib.hc_ksmogn_rsample.restype = None
    lib.hc_cosmos_probs.argtypes = [C.POINTER(_lib.ProbsArgs)]
    lib.hc_cosmos_probs.restype = None
    for n in ("hc_cosmos_sample_globals", "hc_cosmos_sample_locals", "hc_cosmos_elbo_grads",
              "hc_cosmos_globals_grad", "hc_cosmos_adam"):
        getattr(lib, n).argtypes = [C.POINTER(_lib.CosmosArgs)]
        getattr(lib, n).restype = None
    lib.hc_cosmos_adam_catchup.argtypes = [C.POINTER(_lib.CosmosArgs), C.c_int32]
    lib.hc_cosmos_adam_catchup.restype = None
    lib.hc_cosmos_tail_reduced.argtypes = [C.POINTER(_lib.CosmosArgs), C.POINTER(_lib.CosmosArgs)]
    lib.hc_cosmos_tail_reduced.restype = None
    _hc = lib
    return lib


class HostCheckEngine(HipEngine):
    """The engine's host logic (workspace, argument blocks, lazy-Adam clock, sharded step sequence) driving the g++ build
    of the kernels' inline math on host memory instead of libtapqir_hip.so.  Test-side only: it lets the CPU suite run the
    parity comparisons and the world-size-2 gloo test without a GPU.  The product class has no such path."""

    pipelined_tail = False     # the host build has the plain stage functions only
    split_sampling = False
    lazy_adam_default = False

    def _open_library(self):
        return load_hostcheck()

    def struct_sizes(self):
        return int(self.lib.hc_globals_size()), int(self.lib.hc_gbase_size())

    def _interleaved_images(self):
        return None  # the interleaved layout belongs to the GPU kernels

    def _image_stats(self, U):
        self.lib.hc_image_stats(_lib.ptr(self.images), _lib.ptr(self.offset_samples), _lib.ptr(self.pixstats),
                                C.c_int64(U), C.c_int32(self.P))

    def _stream(self):
        return None

    def _blk_floats(self, B):
        return ((B + 255) // 256) * self.n_gsum

    def call(self, name, args):
        getattr(self.lib, "hc_" + name)(C.byref(args))

    def _adam_catchup(self, a, all_units):
        self.lib.hc_cosmos_adam_catchup(C.byref(a), all_units)

    def _tail_reduced(self, a, next_args):
        self.lib.hc_cosmos_tail_reduced(C.byref(a), None if next_args is None else C.byref(next_args))

    def run_probs(self, a):
        self.lib.hc_cosmos_probs(C.byref(a))

    def _run_snr_chi2(self, a):
        self.lib.hc_snr_chi2(C.byref(a))


def CosmosEngine(data, lib=None, **kw):
    """Engine factory of the tests: the HIP engine, or (``lib`` = the loaded host build) its host-check subclass."""
    return HostCheckEngine(data, **kw) if lib is not None else HipEngine(data, **kw)


def make_dataset(N=4, F=6, C=1, P=14, K=2, seed=0, offsets="sim", mask=None):
    """Synthetic data with the reference test-suite parameters (test/test_tapqir.py:20-50)."""
    if C == 1:
        d = simulate(K, N, F, C, P, seed, TEST_PARAMS)
    else:  # independent channels: stack single-channel simulations
        parts = [simulate(K, N, F, 1, P, seed + c, TEST_PARAMS) for c in range(C)]
        d = CosmosDataset(torch.cat([p.images for p in parts], 2), torch.cat([p.xy for p in parts], 2),
                          parts[0].is_ontarget, offset_samples=parts[0].offset.samples,
                          offset_weights=parts[0].offset.weights)
    if offsets == "hist":  # a wide offset histogram like real data (glimpse_reader.py:414-421)
        s = torch.arange(70.0, 110.0)
        w = torch.exp(-0.5 * ((s - 90.0) / 6.0) ** 2)
        d = CosmosDataset(d.images, d.xy, d.is_ontarget, labels=d.labels, offset_samples=s,
                          offset_weights=(w / w.sum()).float())
    if offsets == "peaked":  # weights spanning > 2^40: the packed histogram kernel keeps log-weights inside the exponent
        s = torch.arange(70.0, 110.0)
        w = torch.exp(-0.5 * ((s - 90.0) / 2.0) ** 2).double()
        d = CosmosDataset(d.images, d.xy, d.is_ontarget, labels=d.labels, offset_samples=s, offset_weights=w / w.sum())
    if offsets == "wide":  # offsets reaching above the dimmest pixels: some offsets are masked per pixel (ksmogn.py:226)
        s = torch.arange(70.0, 330.0, 4.0)
        w = torch.exp(-0.5 * ((s - 90.0) / 60.0) ** 2)
        d = CosmosDataset(d.images, d.xy, d.is_ontarget, labels=d.labels, offset_samples=s,
                          offset_weights=(w / w.sum()).float())
    if mask is not None:
        d.mask = mask
    return d


def make_oracle(d, K, perturb=0.3, seed=1, eps=EPS32, crosstalk=False):
    od = OracleData(d.images, d.xy, d.is_ontarget, d.offset.samples, d.offset.weights, mask=d.mask)
    o = (CrosstalkOracle if crosstalk else CosmosOracle)(od, K=K, eps=eps)
    p = o.init_parameters()
    g = torch.Generator().manual_seed(seed)
    for u in p.values():
        if perturb:
            u.data += perturb * torch.randn(u.shape, generator=g, dtype=torch.float64)
        u.data = u.data.float().double()  # identical parameter values on both sides
    return o


def oracle_to_engine(o, eng):
    """Copy the oracle's unconstrained leaves into the engine's flat buffer."""
    views = eng.layout.views(eng.params)
    for n, u in o.params.items():
        views[n].copy_(u.detach().to(eng.params.dtype).reshape(views[n].shape))


def fp32_latents(o, ndx, fdx, seed=3):
    """Native guide draws, rounded to float32 so oracle and kernels consume identical values.
    Returns (lat32 dict of float64 tensors holding fp32 values, base draws consistent with them)."""
    torch.manual_seed(seed)
    with torch.no_grad():
        lat = o.sample_guide(o.params, ndx, fdx)
    lat32 = {k: v.float().double() for k, v in lat.items()}
    lat32["pi"] = torch.stack([1 - lat32["pi"][..., 1], lat32["pi"][..., 1]], -1)
    if "alpha" in lat32:  # the kernels carry both components in fp32; keep them summing to one as drawn
        lat32["alpha"] = torch.stack([lat32["alpha"][..., 0], lat32["alpha"][..., 1]], -1)
    with torch.no_grad():
        dists = o._guide_dists(o.constrained(o.params), ndx, fdx)
        base = o.base_draws(lat32, dists)
    return lat32, base


def put_latents(eng, lat32, base):
    """Write latent draws into the engine's workspace and the global base draws into gbase."""
    K = eng.K
    B = lat32["background"].numel()
    rows = [lat32["background"].reshape(1, B)]
    for name in ("height", "width", "x", "y"):
        rows.append(lat32[name].reshape(K, B))
    eng.lat.copy_(torch.cat(rows, 0).reshape(-1).to(eng.lat.dtype))
    # TqGlobalBase { double gain_g; double prox_t; double lamda_g[4]; double pi_x[4][2]; }
    gb = torch.zeros_like(eng.gbase)
    gb[0] = base["gain_g"]
    gb[1] = base["proximity_t"]
    Q = eng.C
    gb[2:2 + Q] = base["lamda_g"]
    gb[6:6 + 2 * Q] = base["pi_x"].reshape(-1)
    if "alpha_x" in base:  # double alpha_x[2][2] follows pi_x[4][2]
        gb[14:18] = base["alpha_x"].reshape(-1)
    eng.gbase.copy_(gb)


def oracle_grads(o, ndx, fdx, base):
    for u in o.params.values():
        u.grad = None
    lat = o.latents_from_base(o.params, ndx, fdx, base)
    elbo = o.elbo(o.params, ndx, fdx, lat)
    elbo.backward()
    return float(elbo.detach()), {n: (u.grad.clone() if u.grad is not None else torch.zeros_like(u)) for n, u in o.params.items()}


def rel_err(a, b):
    """max |a-b| / max|b| (norm-wise relative error)."""
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-300))


def read_engine_latents(eng, nb, fb):
    """Latent draws of the last step as an oracle-style dict (float64 tensors holding fp32 values)."""
    K, C = eng.K, eng.C
    B = nb * fb * C
    lat = eng.lat.detach().cpu().double().view(1 + 4 * K, B)
    g = eng.globals.detach().cpu().double()
    out = {
        "background": lat[0].view(nb, fb, C),
        "height": lat[1:1 + K].view(K, nb, fb, C),
        "width": lat[1 + K:1 + 2 * K].view(K, nb, fb, C),
        "x": lat[1 + 2 * K:1 + 3 * K].view(K, nb, fb, C),
        "y": lat[1 + 3 * K:1 + 4 * K].view(K, nb, fb, C),
        "gain": g[0].clone(), "proximity": g[1].clone(),
        "lamda": g[5:5 + C].clone(),
        "pi": torch.stack([1 - g[9:9 + C], g[9:9 + C]], -1),
    }
    if getattr(eng, "crosstalk", False):  # float alpha[2][2] follows c[4]
        out["alpha"] = g[21:25].clone().view(2, 2)
    return out
