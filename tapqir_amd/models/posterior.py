"""
Posterior read-out of the cosmos model on the device (tapqir/models/cosmos.py:609-709).

``compute_probs(model)`` returns ``(z_probs, theta_probs)`` with the reference's shapes
``(Nt, F, Q, 1+S)`` and ``(K, Nt, F, Q)``; it launches ``tq_cosmos_probs`` (include/tapqir_hip.h).
"""

import torch

from tapqir_amd import _lib


def probs_args(eng, particles, seed, draw=True, gbase_p=None, xy_given=None):
    """Allocate outputs/workspace for ``tq_cosmos_probs`` and fill the argument block."""
    dev, f32 = eng.device, torch.float32
    U = eng.Nt * eng.F * eng.C
    gsz, bsz = eng.struct_sizes()
    ws = {
        "globals_p": torch.zeros(particles * gsz // 4, dtype=f32, device=dev),
        "gbase_p": torch.zeros(particles * bsz // 8, dtype=torch.float64, device=dev) if gbase_p is None else gbase_p,
        "z_probs": torch.zeros(eng.Nt, eng.F, eng.C, 2, dtype=f32, device=dev),
        "theta_probs": torch.zeros(eng.K, eng.Nt, eng.F, eng.C, dtype=f32, device=dev),
        "xy_given": xy_given,
    }
    a = _lib.ProbsArgs()
    p = _lib.ptr
    a.params, a.is_ontarget = p(eng.params), p(eng.is_ontarget)
    a.globals_p, a.gbase_p, a.xy_given = p(ws["globals_p"]), p(ws["gbase_p"]), p(xy_given)
    a.z_probs, a.theta_probs = p(ws["z_probs"]), p(ws["theta_probs"])
    a.Nt, a.F, a.C, a.P, a.K = eng.Nt, eng.F, eng.C, eng.P, eng.K
    a.particles, a.draw, a.eps, a.seed = particles, int(bool(draw)), eng.eps, seed
    return a, ws


def run_probs(eng, a):
    eng.run_probs(a)


def compute_probs(model, particles=50):
    """cosmos.compute_probs (cosmos.py:609-672): 50 joint guide draws per unit."""
    eng = model.engine
    eng.join()
    a, ws = probs_args(eng, particles, seed=eng.seed + 0x5EED)
    run_probs(eng, a)
    return ws["z_probs"], ws["theta_probs"]
