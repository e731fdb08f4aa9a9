"""Print the worst norm-wise relative gradient error per parity case on the GPU (margin against the 1e-4 bar)."""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import rel_err  # noqa: E402
from test_crosstalk import XT_CASES, XT_IL_CASES, run_xt_case  # noqa: E402
from test_gpu_parity import IL_CASES, run_case_gpu  # noqa: E402
from test_hostcheck_parity import CASES  # noqa: E402


def report(tag, eng, elbo_o, g_o):
    gv = eng.named("grad")
    errs = {n: rel_err(gv[n].cpu().double().reshape(ref.shape), ref) for n, ref in g_o.items()}
    n = max(errs, key=errs.get)
    e = abs(float(eng.elbo_out[0]) - elbo_o) / abs(elbo_o)
    print(f"{tag:48s} elbo {e:.1e}  worst grad {errs[n]:.1e} ({n})", flush=True)


for name, dkw, K, ndx, fdx in CASES:
    o, eng, elbo_o, g_o = run_case_gpu(dkw, K, ndx, fdx)
    report("cosmos/" + name, eng, elbo_o, g_o)
for name, dkw, K in IL_CASES:
    o, eng, elbo_o, g_o = run_case_gpu(dkw, K, None, None, il_min_units=1)
    report("cosmos-il/" + name, eng, elbo_o, g_o)
for name, dkw, K, ndx, fdx in XT_CASES:
    o, eng, elbo_o, g_o = run_xt_case(dkw, K, ndx, fdx, gpu=True)
    report("crosstalk/" + name, eng, elbo_o, g_o)
for name, dkw, K in XT_IL_CASES:
    o, eng, elbo_o, g_o = run_xt_case(dkw, K, None, None, gpu=True, il_min_units=1)
    report("crosstalk-il/" + name, eng, elbo_o, g_o)
