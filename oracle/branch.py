"""Decisions of a forward pass -- ReLU signs and max-pool winners -- recorded or replayed.

The reference's networks are piecewise linear: which linear piece an input lands on is decided by ~1e9 comparisons
(``F.relu``, ``F.max_pool1d``: components.py:41-47, spatial_bb/components.py:34-76,149-168).  An fp32 run and an fp64 run of
the SAME network decide a few dozen of them differently (activations within rounding of zero, pool windows with two nearly
equal entries), and every such decision switches one gradient path: measured on the config-2 step at B = 16, 20 decisions
differ and move the conv weight gradients by 7e-4 .. 3.6e-3 of their peak, while on the SAME branch the fp32 gradients agree
with fp64 to 1e-5 .. 4e-5.  So "fp32 vs fp64" parity of gradients is only meaningful on a common branch: ``Branch`` lets
the oracle replay the decisions another run (the product's) took.

Test infrastructure only -- see ``oracle/__init__.py``.
"""
import torch
from torch.nn import functional as F


class Branch:
    """``Branch()`` records the decisions of the forward it is passed to (``.masks``); ``Branch(masks)`` replays them."""

    def __init__(self, masks=None):
        self.replay = masks is not None
        self.masks = dict(masks) if masks is not None else {}

    def relu(self, x, name):
        if self.replay:
            return x * self.masks[name].to(device=x.device, dtype=x.dtype)      # d/dx = mask: the recorded side of every unit
        self.masks[name] = x > 0
        return F.relu(x)

    def max_pool1d(self, flat, kernel, name):
        """flat [B,1,L] -> [B,1,L//kernel]; the winner of each window is recorded / replayed as an index into L."""
        if self.replay:
            return torch.gather(flat, 2, self.masks[name].to(flat.device))
        out, idx = F.max_pool1d(flat, kernel, return_indices=True)
        self.masks[name] = idx
        return out


class _Plain:
    """No recording: plain F.relu / F.max_pool1d (what the modules do when no branch is given)."""

    @staticmethod
    def relu(x, name):
        return F.relu(x)

    @staticmethod
    def max_pool1d(flat, kernel, name):
        return F.max_pool1d(flat, kernel)


PLAIN = _Plain()
