"""Which regimes of the implicit reparameterisation gradients (tq_math.h: tq_dirichlet_grad_regime, tq_std_gamma_grad) the
draws of a TRAINED fit fall into, per site kind, and how mixed the 64-lane waves / 256-lane workgroups of the sampling
kernel are (consecutive units share a wave).  CPU only; reads gpurun_out/trained_params_<STEPS>.pt (scripts/site_trained.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tapqir_amd.models.layout import ParamLayout

steps = int(os.environ.get("STEPS", 4000))
Nt, F, C, K, P = 400, 1000, 1, 2, 14
lay = ParamLayout(Nt, F, C, K, P, float(torch.finfo(torch.float32).eps))
params = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", f"trained_params_{steps}.pt"), weights_only=True)
v = lay.constrained(params)
torch.manual_seed(0)
U = Nt * F * C

def beta_classes(c1, c0):
    t = torch.distributions.Beta(c1.double(), c0.double()).sample().clamp(1e-38, 1 - 6e-8)
    total = c1 + c0
    out = []
    for x, a, b in ((t, c1, c0), (1 - t, c0, c1)):
        boundary = total * x * (1 - x)
        r = torch.full_like(x, 3, dtype=torch.int64)
        r[(a > 6) & (b > 6)] = 2
        r[(x >= 0.5) & (boundary < 0.75)] = 1
        r[(x <= 0.5) & (boundary < 2.5)] = 0
        out.append(r)
    boundary = total * t * (1 - t)
    pair = (boundary >= 2.5) & (c1 > 6) & (c0 > 6)
    return out[0], out[1], pair

def report(name, c1, c0):
    r0, r1, pair = beta_classes(c1.flatten(), c0.flatten())
    n = r0.numel()
    need = torch.zeros(5, n, dtype=torch.bool)  # classes: P(pair), 0, 1, 2(single), 3
    need[0] = pair
    for r in (r0, r1):
        for k in range(4):
            need[1 + k] |= (r == k) & ~pair
    frac = need.float().mean(1)
    w = need[:, : n // 64 * 64].view(5, -1, 64).any(2).float().mean(1)
    g = need[:, : n // 256 * 256].view(5, -1, 256).any(2).float().mean(1)
    cnt256 = need[:, : n // 256 * 256].view(5, -1, 256).sum(2).float()
    waves_compacted = ((cnt256 + 63) // 64).mean(1) / 4  # waves (of 4) that would run the class after compaction in the workgroup
    print(f"{name}: lanes needing [pair, x-small, (1-x)-small, single-saddle, rational] = {[round(float(x), 3) for x in frac]}")
    print(f"    waves (64 lanes) that execute the class now: {[round(float(x), 3) for x in w]}")
    print(f"    workgroups (256) with any lane in the class: {[round(float(x), 3) for x in g]};  after compaction, fraction of waves running it: {[round(float(x), 3) for x in waves_compacted]}")

H = (P + 1) / 2
for k in range(K):
    size = v["size"][k].flatten()
    for nm in ("x_mean", "y_mean"):
        mean = v[nm][k].flatten()
        c1 = size * (mean + H) / (2 * H)
        report(f"{nm}[{k}]", c1, size - c1)
    ws = v["w_size"][k].flatten()
    wm = v["w_mean"][k].flatten()
    c1 = ws * (wm - 0.75) / 1.5
    report(f"w[{k}]", c1, ws - c1)

def gamma_report(name, loc, beta):
    alpha = (loc * beta).flatten().double()
    x = torch.distributions.Gamma(alpha, torch.ones_like(alpha)).sample().clamp_min(1e-38)
    cls = torch.full_like(x, 2, dtype=torch.int64)
    cls[alpha > 8] = 1
    cls[x < 0.8] = 0
    need = torch.stack([cls == k for k in range(3)])
    n = x.numel()
    frac = need.float().mean(1)
    w = need[:, : n // 64 * 64].view(3, -1, 64).any(2).float().mean(1)
    cnt256 = need[:, : n // 256 * 256].view(3, -1, 256).sum(2).float()
    print(f"{name}: lanes [taylor x<0.8, saddle alpha>8, rational] = {[round(float(x), 3) for x in frac]}; waves now {[round(float(x), 3) for x in w]}; "
          f"after compaction {[round(float(x), 3) for x in ((cnt256 + 63) // 64).mean(1) / 4]}; alpha<1 (boosted draw): {float((alpha < 1).float().mean()):.3f}")

gamma_report("b", v["b_loc"], v["b_beta"])
for k in range(K):
    gamma_report(f"h[{k}]", v["h_loc"][k], v["h_beta"][k])
