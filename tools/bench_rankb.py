#!/usr/bin/env python3
"""dd_adam_step_rankb alone against dd_linear_wgrad + dd_adam_step on the three big Linear weights (encoder fc1 940032 -> 128, head
64 -> 640000, decoder fc2 128 -> 1253376), batch 32: time, bytes, and the difference of the two results.
DD_RANKB_VARIANT (diagnostic builds only) picks a kernel variant; one process per variant (the choice is read once)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from driving_dirty_amd import ops  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

dev = torch.device("cuda:0")
shapes = {"fc1": (128, 940032), "head": (640000, 64), "dec_fc2": (1253376, 128), "fc1_2x": (128, 4080128)}      # fc1_2x: config 5 (ROWS=16)
rows = int(os.environ.get("ROWS", "32"))
if os.environ.get("BLOCKS"):      # persistent workgroups per CU of the optimizer kernels (TrainStep: 1 beside the backward, 4 after it)
    ops.check(ops._lib.lib().dd_set_adam_blocks_per_cu(int(os.environ["BLOCKS"])), "dd_set_adam_blocks_per_cu")
for name, (n, k) in shapes.items():
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    torch.manual_seed(1)
    p0 = torch.randn(n, k, device=dev) * 0.02
    x = torch.randn(rows, k, device=dev)
    dy = torch.randn(rows, n, device=dev) * 1e-3
    b0 = torch.randn(n, device=dev) * 0.02
    # materialised path
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    b, bm, bv = b0.clone(), torch.zeros_like(b0), torch.zeros_like(b0)
    dw, db = torch.empty_like(p0), torch.empty_like(b0)

    def old(step=1):
        ops.check(ops._lib.lib().dd_linear_wgrad(ops._p(dy), ops._p(x), ops._p(dw), ops._p(db), rows, n, k, ops._stream()), "wgrad")
        ops.adam_step_flat(p.view(-1), dw.view(-1), m.view(-1), v.view(-1), 1e-3, 0.9, 0.999, 1e-8, step, 1.0)
        ops.adam_step_flat(b, db, bm, bv, 1e-3, 0.9, 0.999, 1e-8, step, 1.0)

    p2, m2, v2 = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    b2, bm2, bv2 = b0.clone(), torch.zeros_like(b0), torch.zeros_like(b0)

    def new(step=1):
        ops.adam_step_rankb(p2, m2, v2, dy, x, b2, bm2, bv2, 1e-3, 0.9, 0.999, 1e-8, step, 1.0)

    for s in (1, 2):
        old(s)
        new(s)
    torch.cuda.synchronize()
    dp = float((p - p2).abs().max())
    dm = float((m - m2).abs().max() / m.abs().max())
    dv = float((v - v2).abs().max() / v.abs().max())
    dbias = float((b - b2).abs().max())
    t_old = timeit(lambda: old(3), 10)
    t_new = timeit(lambda: new(3), 10)
    nb = n * k * 4
    print(f"{name}: wgrad+adam {t_old:.3f} ms ({(8 * nb + rows * (n + k) * 4) / t_old / 1e9:.2f} TB/s)   rankb {t_new:.3f} ms "
          f"({(6 * nb + rows * (n + k) * 4) / t_new / 1e9:.2f} TB/s)   |dp| {dp:.2e} (lr 1e-3)  dm {dm:.1e} dv {dv:.1e} |dbias| {dbias:.1e}", flush=True)
    del p, m, v, p2, m2, v2, dw, p0
    torch.cuda.empty_cache()
