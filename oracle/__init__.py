"""
CPU oracle for the cosmos SVI hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, float64) restatement of the reference's
algorithm for the path named by BASELINE.json's ``north_star``.  It exists so
that the hand-written HIP kernels in ``tapqir_amd/csrc`` can be checked against
an independent implementation.  Nothing in ``tapqir_amd`` imports it; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may.

Pinning status
--------------
* ``oracle.dist_util`` (gaussian_spots, truncated_poisson_probs, probs_m,
  probs_theta, expand_offtarget) is PINNED: it is checked at float64 against
  vectors produced by importing the reference's own
  ``tapqir/distributions/util.py`` in the build container
  (``tests/golden/make_golden.py`` -> ``tests/golden/util_golden.npz``).
* per-site densities are checked against the installed ``torch.distributions``
  objects, and the KSMOGN mixture against a brute-force per-pixel,
  per-offset ``Gamma.log_prob`` loop.
* The ELBO itself (Pyro ``TraceEnum_ELBO`` semantics: Dice weights over the
  guide-enumerated ``m``, exact marginalisation of the model-enumerated
  ``z, theta``, plate scaling, masks) is **PARITY UNPINNED**: pyro-ppl,
  funsor and pykeops are not installed in this image, the reference's tests
  hold no numerical assertions for this path (``test/test_tapqir.py:91-93``
  checks the exit code only), so the ELBO restatement is pinned only by an
  independent brute-force enumerator (``oracle.cosmos.elbo_bruteforce``)
  written from the model's published joint density
  (``tapqir/models/cosmos.py:139-167``).
"""
