"""
Dataset container and ``data.tpqr`` file format of the drop-in surface.

Mirrors tapqir/utils/dataset.py:18-222 (``OffsetData``, ``CosmosDataset``, ``save``,
``load``): same attribute names, same dict keys in the file, so a ``data.tpqr`` written by
the reference loads here and vice versa.  ``Vindex`` is replaced by plain advanced
indexing (the only use, ``fetch``, indexes with broadcastable index tensors).
"""

import logging
from collections import namedtuple
from pathlib import Path

import torch
from torch.distributions.utils import lazy_property, probs_to_logits

from tapqir_amd.exceptions import TapqirFileNotFoundError

logger = logging.getLogger(__name__)


class OffsetData(namedtuple("OffsetData", ["samples", "weights"])):
    """Empirical camera-offset distribution (dataset.py:18-37)."""

    @lazy_property
    def min(self):
        return torch.min(self.samples).item()

    @lazy_property
    def max(self):
        return torch.max(self.samples).item()

    @lazy_property
    def logits(self):
        return probs_to_logits(self.weights)

    @lazy_property
    def mean(self):
        return torch.sum(self.samples * self.weights).item()

    @lazy_property
    def var(self):
        return torch.sum(self.samples**2 * self.weights).item() - self.mean**2


class CosmosDataset:
    """images (Nt,F,C,P,P), xy (Nt,F,C,2), is_ontarget (Nt,), mask (Nt,), labels, offset
    (dataset.py:40-192)."""

    def __init__(self, images, xy, is_ontarget, mask=None, labels=None, offset_samples=None,
                 offset_weights=None, device=torch.device("cpu"), time1=None, ttb=None,
                 name=None, channels=None):
        self.images = images
        self.xy = xy
        self.is_ontarget = is_ontarget
        if mask is None:
            mask = torch.ones_like(is_ontarget, dtype=torch.bool)
        self.mask = mask
        self.labels = labels
        self.device = device
        self.offset = OffsetData(offset_samples.to(device), offset_weights.to(device))
        self.time1 = time1
        self.ttb = ttb
        self.name = name
        if channels is None:
            channels = tuple(f"channel{c}" for c in range(self.C))
        self.channels = channels

    @lazy_property
    def N(self) -> int:
        return int(self.is_ontarget.sum().item())

    @lazy_property
    def Nc(self) -> int:
        return int((~self.is_ontarget).sum().item())

    @lazy_property
    def Nt(self) -> int:
        return self.N + self.Nc

    @property
    def F(self) -> int:
        return self.images.shape[1]

    @property
    def C(self) -> int:
        return self.images.shape[2]

    @property
    def P(self) -> int:
        Px, Py = self.images.shape[3], self.images.shape[4]
        assert Px == Py
        return Px

    @property
    def x(self) -> torch.Tensor:
        return self.xy[..., 0]

    @property
    def y(self) -> torch.Tensor:
        return self.xy[..., 1]

    @lazy_property
    def median(self) -> torch.Tensor:
        return torch.stack([torch.median(self.images[..., c, :, :]) for c in range(self.C)])

    def fetch(self, ndx, fdx, cdx):
        """dataset.py:140-151: gather a minibatch and move it to the compute device."""
        cpu = lambda i: i.cpu() if isinstance(i, torch.Tensor) and self.images.device.type == "cpu" else i
        ndx, fdx, cdx = cpu(ndx), cpu(fdx), cpu(cdx)
        return (
            self.images[ndx, fdx, cdx].to(self.device),
            self.xy[ndx, fdx, cdx].to(self.device),
            self.is_ontarget[ndx].to(self.device),
        )

    def _quantile(self, q):
        return torch.stack([
            torch.quantile(self.images[..., c, :, :].flatten().float()[:16_000_000], q) for c in range(self.C)
        ])

    @lazy_property
    def vmin(self) -> torch.Tensor:
        return self._quantile(0.05)

    @lazy_property
    def vmax(self) -> torch.Tensor:
        return self._quantile(0.99)

    def __repr__(self):
        return (f"{self.__class__.__name__}: {self.name}"
                f"\n  images           tensor(N={self.N} on-target AOIs, Nc={self.Nc} off-target AOIs, "
                f"F={self.F} frames, C={self.C} channels, P={self.P} pixels, P={self.P} pixels)"
                f"\n  offset.samples   {self.offset.samples!r}"
                f"\n        .weights   {self.offset.weights!r}")


def save(obj, path):
    """dataset.py:195-212 -- same keys."""
    path = Path(path)
    cpu = lambda t: t.cpu() if isinstance(t, torch.Tensor) else t
    torch.save(
        {
            "images": cpu(obj.images), "xy": cpu(obj.xy), "is_ontarget": cpu(obj.is_ontarget),
            "mask": cpu(obj.mask), "labels": obj.labels,
            "offset_samples": cpu(obj.offset.samples), "offset_weights": cpu(obj.offset.weights),
            "name": obj.name, "time1": obj.time1, "ttb": obj.ttb, "channels": obj.channels,
        },
        path / "data.tpqr",
    )
    logger.info(f"Data is saved in {path / 'data.tpqr'}")


def load(path, device=torch.device("cpu")):
    """dataset.py:215-222."""
    path = Path(path)
    try:
        data_tapqir = torch.load(path / "data.tpqr", weights_only=False)
    except FileNotFoundError:
        raise TapqirFileNotFoundError("data", path / "data.tpqr")
    return CosmosDataset(**data_tapqir, **{"device": device})
