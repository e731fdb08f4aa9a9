"""Model registry of the drop-in surface (tapqir/models/__init__.py:4-21)."""

from tapqir_amd.models.cosmos import Cosmos, cosmos
from tapqir_amd.models.model import Model

__all__ = ["models", "Model", "cosmos", "Cosmos"]

models = {
    cosmos.name: cosmos,
}
