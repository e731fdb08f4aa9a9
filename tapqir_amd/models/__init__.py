"""Model registry of the drop-in surface (tapqir/models/__init__.py:4-21)."""

from tapqir_amd.models.cosmos import Cosmos, cosmos
from tapqir_amd.models.crosstalk import crosstalk
from tapqir_amd.models.model import Model


class hmm(cosmos):
    """``cosmos+hmm`` (tapqir/models/hmm.py) is outside the hot-path scope of this build:
    the name is registered so that callers get a clear error instead of a KeyError."""

    name = "cosmos+hmm"

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("the cosmos+hmm model (funsor-based) is not implemented in tapqir_amd; "
                                  "use 'cosmos' or 'crosstalk'")


__all__ = ["models", "Model", "cosmos", "Cosmos", "crosstalk", "hmm"]

models = {
    cosmos.name: cosmos,
    crosstalk.name: crosstalk,
    hmm.name: hmm,
}
