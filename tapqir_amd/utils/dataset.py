"""
Dataset container and ``data.tpqr`` file format of the drop-in surface.

Public surface of tapqir/utils/dataset.py:18-222 -- ``OffsetData`` (samples, weights + min / max / logits / mean /
var), ``CosmosDataset`` (images, xy, is_ontarget, mask, labels, offset, time1, ttb, name, channels; sizes N, Nc, Nt,
F, C, P; x, y, median, vmin, vmax, fetch), ``save`` and ``load`` with the file's dict keys unchanged -- so a
``data.tpqr`` written by the reference loads here and vice versa.  ``fetch`` uses plain advanced indexing (its index
tensors are broadcastable; no ``Vindex`` needed).
"""

import logging
from pathlib import Path
from typing import NamedTuple

import torch
from torch.distributions.utils import probs_to_logits

from tapqir_amd.exceptions import TapqirFileNotFoundError

logger = logging.getLogger(__name__)

DATA_FILE = "data.tpqr"
# keys of the saved dict, in file order (dataset.py:195-212)
FILE_KEYS = ("images", "xy", "is_ontarget", "mask", "labels", "offset_samples", "offset_weights", "name", "time1", "ttb",
             "channels")


class OffsetData(NamedTuple):
    """Empirical camera-offset distribution: support points and their probabilities (dataset.py:18-37)."""

    samples: torch.Tensor
    weights: torch.Tensor

    def _moment(self, power):
        return float((self.samples ** power * self.weights).sum())

    @property
    def min(self):
        return self.samples.min().item()

    @property
    def max(self):
        return self.samples.max().item()

    @property
    def logits(self):
        return probs_to_logits(self.weights)

    @property
    def mean(self):
        return self._moment(1)

    @property
    def var(self):
        return self._moment(2) - self._moment(1) ** 2


class CosmosDataset:
    """AOI images ``(Nt, F, C, P, P)`` with target positions ``xy (Nt, F, C, 2)``, the on-target flag and fit mask per
    AOI, optional labels and acquisition times, and the offset distribution (dataset.py:40-192)."""

    def __init__(self, images, xy, is_ontarget, mask=None, labels=None, offset_samples=None, offset_weights=None,
                 device=torch.device("cpu"), time1=None, ttb=None, name=None, channels=None):
        self.images, self.xy, self.is_ontarget = images, xy, is_ontarget
        self.mask = torch.ones_like(is_ontarget, dtype=torch.bool) if mask is None else mask
        self.labels, self.time1, self.ttb, self.name = labels, time1, ttb, name
        self.device = device
        self.offset = OffsetData(offset_samples.to(device), offset_weights.to(device))
        self.channels = tuple(f"channel{c}" for c in range(images.shape[2])) if channels is None else channels
        if images.shape[3] != images.shape[4]:
            raise ValueError(f"AOI images must be square, got {tuple(images.shape[3:5])}")
        self._cache = {}

    # -- sizes ----------------------------------------------------------------------------------------------
    def _cached(self, key, fn):
        if key not in self._cache:
            self._cache[key] = fn()
        return self._cache[key]

    N = property(lambda self: self._cached("N", lambda: int(self.is_ontarget.sum())), doc="number of on-target AOIs")
    Nc = property(lambda self: self._cached("Nc", lambda: int((~self.is_ontarget).sum())), doc="number of off-target AOIs")
    Nt = property(lambda self: self.N + self.Nc, doc="total number of AOIs")
    F = property(lambda self: self.images.shape[1], doc="number of frames")
    C = property(lambda self: self.images.shape[2], doc="number of colour channels")
    P = property(lambda self: self.images.shape[3], doc="AOI side in pixels")
    x = property(lambda self: self.xy[..., 0], doc="target position along the column axis")
    y = property(lambda self: self.xy[..., 1], doc="target position along the row axis")

    # -- intensity summaries (per channel) ----------------------------------------------------------------------
    def _per_channel(self, fn):
        return torch.stack([fn(self.images[:, :, c]) for c in range(self.C)])

    @property
    def median(self) -> torch.Tensor:
        return self._cached("median", lambda: self._per_channel(torch.median))

    def _quantile(self, q):
        # torch.quantile is limited to 16 M elements: the head of the data is representative for display ranges
        return self._per_channel(lambda im: torch.quantile(im.flatten().float()[:16_000_000], q))

    @property
    def vmin(self) -> torch.Tensor:
        return self._cached("vmin", lambda: self._quantile(0.05))

    @property
    def vmax(self) -> torch.Tensor:
        return self._cached("vmax", lambda: self._quantile(0.99))

    # -- minibatch gather (dataset.py:140-151) -----------------------------------------------------------------
    def fetch(self, ndx, fdx, cdx):
        """Images, target positions and on-target flags of a minibatch, on the compute device."""
        if self.images.device.type == "cpu":
            ndx, fdx, cdx = (i.cpu() if isinstance(i, torch.Tensor) else i for i in (ndx, fdx, cdx))
        dev = self.device
        return self.images[ndx, fdx, cdx].to(dev), self.xy[ndx, fdx, cdx].to(dev), self.is_ontarget[ndx].to(dev)

    def __repr__(self):
        return (f"{type(self).__name__}: {self.name}\n"
                f"  images           tensor(N={self.N} on-target AOIs, Nc={self.Nc} off-target AOIs, F={self.F} frames, "
                f"C={self.C} channels, P={self.P} pixels, P={self.P} pixels)\n"
                f"  offset.samples   {self.offset.samples!r}\n"
                f"        .weights   {self.offset.weights!r}")


def save(obj, path):
    """Write ``<path>/data.tpqr`` with the reference's keys (dataset.py:195-212)."""
    target = Path(path) / DATA_FILE
    values = {"offset_samples": obj.offset.samples, "offset_weights": obj.offset.weights}
    payload = {}
    for key in FILE_KEYS:
        value = values[key] if key in values else getattr(obj, key)
        payload[key] = value.cpu() if isinstance(value, torch.Tensor) else value
    torch.save(payload, target)
    logger.info(f"Data is saved in {target}")


def load(path, device=torch.device("cpu")):
    """Read ``<path>/data.tpqr`` (dataset.py:215-222); a missing file raises TapqirFileNotFoundError."""
    source = Path(path) / DATA_FILE
    if not source.is_file():
        raise TapqirFileNotFoundError("data", source)
    from tapqir_amd.utils.safe_load import load_tpqr

    return CosmosDataset(**load_tpqr(source), device=device)
