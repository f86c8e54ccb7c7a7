"""Three-way comparison of a product run with the fp64 oracle and a reference-generated fixture, for piecewise-linear
networks (oracle/branch.py).

    product - fixture  =  (product - oracle on the PRODUCT'S branch)     arithmetic of the kernels: held to 2e-4 of peak
                        + (oracle on the product's branch - oracle FREE)  effect of the decisions that differ: measured, and
                                                                          certified by the census to be rounding-level flips
                        + (oracle free - fixture)                         the oracle run here IS the reference's fp64 run: 1e-8

so no blanket cross-branch tolerance is needed: each tensor's budget against the fixture is
``max(1e-3, 2 x the reference's own fp32-vs-fp64 deviation on that tensor, measured branch effect + 2e-4)``, and the census
(``oracle.branch.Census``) asserts that the product's ReLU / max-pool decisions differ from the free fp64 run's only at units
whose fp64 pre-activation is within rounding of zero -- a kernel that wrongly zeroes part of a ReLU layer fails there.

Test infrastructure (imports oracle/)."""
import json
import os

import numpy as np
import torch

TOL = 1e-3
SAME_BRANCH_TOL = 2e-4
PIN_TOL = 1e-8
DUMP = os.environ.get("DD_DUMP_CENSUS", "")


def rel_err(got, ref, floor=1e-30):
    got, ref = torch.as_tensor(got).detach().double().cpu(), torch.as_tensor(ref).detach().double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(floor))


def samp(t, idx):
    return t.detach().reshape(-1)[torch.from_numpy(np.asarray(idx)).to(t.device)]


def fixture_entry(g, name, prefix="", sample_prefix=None):
    """(f64, f32, idx or None) of tensor ``name`` in fixture ``g``: stored whole as ``grad.<name>`` or sampled as
    ``gradsamp.<name>`` + ``gradidx.<name>``; None when the fixture does not hold it."""
    full = f"grad.{prefix}{name}"
    if full + "_f64" in g.files:
        return g[full + "_f64"], g[full + "_f32"], None
    s = f"gradsamp.{prefix}{name}"
    if s + "_f64" in g.files:
        return g[s + "_f64"], g[s + "_f32"], g[f"gradidx.{prefix}{name}"]
    return None


def three_way(label, product, run_oracle, masks, fixture=None, floors=None, census_args=None, same_tol=SAME_BRANCH_TOL,
              pin_tol=PIN_TOL, exact=("loss",), run_oracle32=None):
    """product: name -> tensor (the HIP path's results).  run_oracle(branch) -> name -> fp64 tensor, evaluated with the given
    ``oracle.branch`` object.  masks: the product's decisions (name -> bool mask / pool index).  fixture: name -> (f64, f32, idx)
    from the reference-generated file (``fixture_entry``), for the tensors it holds.  floors: name -> denominator floor for tensors
    that are zero by construction.  run_oracle32 (ill-conditioned cases only: train-mode BatchNorm1d over a handful of rows):
    the same oracle in fp32 -- plain torch arithmetic, the reference's own -- replaying the same branch; a tensor may then be as
    far from the fp64 result ON THE SAME BRANCH as twice what that fp32 run is (measured per tensor; 2e-4 otherwise).
    Returns the table it asserted on."""
    from oracle.branch import Branch, Census
    fixture, floors = fixture or {}, floors or {}
    census = Census(masks)
    free = {k: v.detach().clone() for k, v in run_oracle(census).items()}
    report = census.check(**(census_args or {}))
    same = {k: v.detach().clone() for k, v in run_oracle(Branch(masks)).items()}
    same32 = {k: v.detach().clone() for k, v in run_oracle32(Branch(masks)).items()} if run_oracle32 is not None else None
    table, bad = {}, {}
    for k, got in product.items():
        fl = floors.get(k, 1e-30)
        row = {"same_branch": rel_err(got, same[k], fl), "branch_effect": rel_err(same[k], free[k], fl)}
        tol = 1e-6 if k in exact else same_tol
        if same32 is not None and k not in exact:
            row["torch_fp32_same_branch"] = rel_err(same32[k], same[k], fl)
            tol = max(tol, 2.0 * row["torch_fp32_same_branch"])
        row["same_branch_tol"] = tol
        if not row["same_branch"] < tol:
            bad[k] = dict(row, why=f"beyond {tol:g} of peak on the product's own branch")
        if k in fixture:
            f64, f32, idx = fixture[k]
            pick = (lambda t: samp(t, idx)) if idx is not None else (lambda t: t)
            row["oracle_vs_fixture"] = rel_err(pick(free[k]), f64, fl)
            row["ref_fp32_vs_fp64"] = float(np.abs(f64 - f32).max() / max(np.abs(f64).max(), fl))
            row["vs_fixture"] = rel_err(pick(torch.as_tensor(got)), f64, fl)
            # the effect of the differing decisions on exactly what the fixture holds (its sample, its normalisation)
            effect = rel_err(pick(same[k]), pick(free[k]), fl) if idx is not None else row["branch_effect"]
            if idx is not None:
                row["branch_effect_on_sample"] = effect
            row["budget"] = max(TOL, 2.0 * row["ref_fp32_vs_fp64"], 1.05 * effect + tol) if k not in exact else 1e-5
            if not row["oracle_vs_fixture"] < pin_tol:
                bad[k] = dict(row, why="the fp64 oracle run here differs from the reference-generated fixture")
            elif not row["vs_fixture"] < row["budget"]:
                bad[k] = dict(row, why="beyond the per-tensor budget against the reference-generated fixture")
        table[k] = row
    if DUMP:
        os.makedirs(os.path.dirname(DUMP) or ".", exist_ok=True)
        with open(DUMP, "a") as f:
            f.write(json.dumps({"test": label, "census": report, "tensors": table}) + "\n")
    assert not bad, f"{label}: {bad}\ncensus: {report}\nall: {table}"
    return table, report


# ------------------------------------------------------------------------------------------------ the product's decisions
def _nchw(t):
    return t.permute(0, 3, 1, 2)


def encoder_masks(tr, dense_at=0, pool=True):
    """ops.TRACE of one EncoderConvStack pass (+ its two DenseBlocks at tr['dense'][dense_at:dense_at+2]) -> oracle names."""
    import torch.nn.functional as F
    a3 = _nchw(tr["a3"]).contiguous()
    m = {"relu1": _nchw(tr["a1"]) > 0, "relu2": _nchw(tr["a2"]) > 0, "relu3": a3 > 0}
    if pool:
        m["pool"] = F.max_pool1d(a3.reshape(a3.size(0), 1, -1), 4, return_indices=True)[1]
        m["fc1"], m["fc2"] = tr["dense"][dense_at] > 0, tr["dense"][dense_at + 1] > 0      # dropout is off: y > 0 iff the ReLU passed
    return m


def decoder_masks(tr, dense_at=0):
    return {"d_fc1": tr["dense"][dense_at] > 0, "d_fc2": tr["dense"][dense_at + 1] > 0,
            "dc1": _nchw(tr["dc1"]) > 0, "dc2": _nchw(tr["dc2"]) > 0, "dc3": _nchw(tr["dc3"]) > 0}


def merge_masks(tr, with_rm):
    m = {"ss_conv": _nchw(tr["s1"]) > 0, "ss_deconv": _nchw(tr["cat"][..., 0:32]) > 0}
    if with_rm:
        m.update(rm1=_nchw(tr["r1"]) > 0, rm2=_nchw(tr["cat"][..., 64:96]) > 0)
    for i, a in enumerate(tr["acts"][1:]):
        m[f"up{i + 1}"] = _nchw(a) > 0
    return m


def spatial_masks(tr):
    th, tw = tr["tile"]
    m = {"out": _nchw(tr["space_out"]) > 0}
    for name, (r, c) in (("bl", (0, 0)), ("fl", (0, 1)), ("b", (1, 0)), ("f", (1, 1)), ("br", (2, 0)), ("fr", (2, 1))):
        m[name] = _nchw(tr["mosaic"][:, r * th:(r + 1) * th, c * tw:(c + 1) * tw, :]) > 0
    return m


def grads_of(module, prefix=""):
    return {prefix + k: p.grad.detach().clone() for k, p in module.named_parameters() if p.grad is not None}


def bn_bias_floors(grads, names):
    """A Linear / conv bias in front of a train-mode BatchNorm has an exactly-zero gradient: its fp32 value is rounding noise
    of the layer's gradient scale and is judged against the peak of that layer's WEIGHT gradient."""
    return {k: float(grads[k[:-4] + "weight"].abs().max()) for k in names if k in grads}
