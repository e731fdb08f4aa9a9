"""
Loading of ``.tpqr`` files (``torch.save`` pickles: data.tpqr, <name>_model.tpqr, <name>_params.tpqr) without executing
anything from the file: ``torch.load(weights_only=True)`` plus an allow-list of the few non-tensor classes the
reference's payloads contain (tapqir/utils/dataset.py:195-212, tapqir/models/model.py:272-282): numpy arrays (labels
are structured arrays), ``collections.deque`` (the rolling convergence window) and the
``torch.distributions.constraints`` classes of the parameter store.
"""

import collections

import numpy as np
import torch
from torch.distributions import constraints


def _allowed():
    out = [collections.deque, collections.defaultdict, collections.OrderedDict, np.ndarray, np.dtype]
    try:
        from numpy._core.multiarray import _reconstruct, scalar
    except ImportError:  # numpy < 2
        from numpy.core.multiarray import _reconstruct, scalar
    out += [_reconstruct, scalar]
    # dtype classes of structured label arrays and plain numeric arrays
    out += [type(np.dtype(t)) for t in ("int64", "int32", "float64", "float32", "bool", "uint8", "int16", "uint16")]
    out += [type(np.dtype([("aoi", int)]))]
    # constraint classes (instances such as `positive` pickle as their class + state)
    for name in dir(constraints):
        obj = getattr(constraints, name)
        if isinstance(obj, type) and issubclass(obj, constraints.Constraint):
            out.append(obj)
        elif isinstance(obj, constraints.Constraint):
            out.append(type(obj))
    seen, uniq = set(), []
    for o in out:
        if id(o) not in seen:
            seen.add(id(o))
            uniq.append(o)
    return uniq


def load_tpqr(path, map_location=None):
    """``torch.load`` restricted to tensors, containers and the allow-listed classes above; anything else in the file
    raises ``pickle.UnpicklingError``.  Checkpoints written by the reference itself hold ``collections.deque`` windows
    (model.py:279), which the restricted unpickler cannot rebuild, and may hold pyro classes: such a file is refused
    unless the user vouches for it with ``TAPQIR_AMD_TRUST_FILES=1`` (full unpickling, runs code from the file)."""
    import os
    import pickle

    try:
        with torch.serialization.safe_globals(_allowed()):
            return torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as err:
        if os.environ.get("TAPQIR_AMD_TRUST_FILES") == "1":
            return torch.load(path, map_location=map_location, weights_only=False)
        raise pickle.UnpicklingError(
            f"{path}: refused by the restricted loader ({str(err).splitlines()[-1] if str(err) else err}). "
            "Files written by tapqir_amd load without it; set TAPQIR_AMD_TRUST_FILES=1 to unpickle a file you trust "
            "(e.g. a checkpoint written by the reference, which stores deque windows).") from None
