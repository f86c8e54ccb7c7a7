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


class Census:
    """A FREE-running branch (its own F.relu / F.max_pool1d decisions, like ``PLAIN``) that compares every decision with the
    ones another run recorded (``theirs``: name -> bool mask / winner index, e.g. taken from the product's activations) and
    keeps the tally in ``report``.  This is what makes the same-branch comparisons checkable: replaying the product's
    decisions proves the arithmetic on the product's linear piece; the census proves that piece is the right one -- the two
    runs may only differ at units whose fp64 pre-activation is within rounding of zero (pool: windows whose two candidates
    are within rounding of each other).  A kernel that wrongly zeroes a region of a ReLU layer shows up here as units that
    differ at a large fp64 pre-activation.

    report[name] = {"kind", "units", "differ", "worst", "peak"}: ``worst`` is the largest |fp64 pre-activation| among the
    differing ReLU units (pool: the largest gap between this run's winner and the other run's winner), ``peak`` the
    largest |pre-activation| of the layer (pool: the largest |entry|)."""
    replay = False

    def __init__(self, theirs):
        self.theirs = theirs
        self.report = {}
        self.masks = {}

    def relu(self, x, name):
        own = x > 0
        diff = own != self.theirs[name].to(device=x.device)
        n = int(diff.sum())
        xd = x.detach()
        self.report[name] = {"kind": "relu", "units": x.numel(), "differ": n, "peak": float(xd.abs().max()),
                             "worst": float(xd[diff].abs().max()) if n else 0.0}
        self.masks[name] = own
        return F.relu(x)

    def max_pool1d(self, flat, kernel, name):
        out, idx = F.max_pool1d(flat, kernel, return_indices=True)
        theirs = self.theirs[name].to(flat.device)
        diff = idx != theirs
        n = int(diff.sum())
        gap = (out - torch.gather(flat, 2, theirs)).detach()
        self.report[name] = {"kind": "pool", "units": idx.numel(), "differ": n, "peak": float(flat.detach().abs().max()),
                             "worst": float(gap[diff].abs().max()) if n else 0.0}
        self.masks[name] = idx
        return out

    def check(self, max_frac=1e-5, eps=2e-5, per_layer=None):
        """Every layer: no unit decided differently at more than ``eps`` x the layer's peak (pool: no window whose two winners
        are further apart than that), and -- ReLU layers -- at most ``max_frac`` of the units (never fewer than a handful
        allowed) decided differently at all.  Pool windows with tied entries (a constant region: the blanked view of the
        masked-view task, all-zero windows behind a ReLU) pick either winner; their count is reported, not bounded.
        ``per_layer``: name -> (max_frac, eps) overrides.  Returns the report; raises AssertionError naming the offenders."""
        bad = {}
        for name, r in self.report.items():
            mf, e = (per_layer or {}).get(name, (max_frac, eps))
            allowed = max(4, int(mf * r["units"]))
            too_many = r["kind"] == "relu" and r["differ"] > allowed
            if too_many or r["worst"] > e * max(r["peak"], 1e-300):
                bad[name] = dict(r, allowed=allowed, eps_abs=e * r["peak"])
        assert not bad, f"decisions differ beyond rounding: {bad}\nall: {self.report}"
        return self.report


class _Plain:
    """No recording: plain F.relu / F.max_pool1d (what the modules do when no branch is given)."""

    @staticmethod
    def relu(x, name):
        return F.relu(x)

    @staticmethod
    def max_pool1d(flat, kernel, name):
        return F.max_pool1d(flat, kernel)


PLAIN = _Plain()
