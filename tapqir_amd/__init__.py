"""tapqir_amd -- the cosmos / crosstalk SVI hot path of Tapqir on AMD MI355X."""

__version__ = "0.1.0"
