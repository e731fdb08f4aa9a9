"""Glimpse / imscroll input side (tapqir/imscroll/__init__.py)."""

from tapqir_amd.imscroll.glimpse_reader import GlimpseDataset, bin_hist, read_glimpse

__all__ = ["GlimpseDataset", "bin_hist", "read_glimpse"]
