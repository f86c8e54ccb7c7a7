#!/usr/bin/env python3
"""Headline benchmark: 6-view scenes/sec, roadmap model, fwd + bwd + Adam, bs = 32 per GPU, fp32.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; ranks shard the batch (weak scaling: 32 scenes per GPU), gradients are all-reduced
over RCCL (driving_dirty_amd.ddp.GradSync) overlapped with the conv backward.  A "step" is one pass of the
hot path over one synthetic batch: 6-view stitch -> conv encoder -> pool -> dense blocks -> Linear(64, 640000)
-> BCE-with-logits, backward through everything, Adam on all 162 M parameters.  Inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line (fields: see the task contract / DESIGN.md section 6).

Workload = BASELINE.json configs[1] (reference src/roadmap_model/roadmap_bce_v2.py with the report's best AE:
hidden 128 / latent 64, FinalReport Table 1), synthetic 6x3x256x306 images and 800x800 road masks, default
PyTorch init under the reference's seed 20200505.
"""
import argparse
import json
import os
import sys
import time
from argparse import Namespace

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20200505                     # reference autoencoder.py:16-18
BATCH, H, W = 32, 256, 306
HIDDEN, LATENT = 128, 64
PEAK_F32_MFMA_TF = 157.3            # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix)
# algorithmic work per scene, fwd+bwd: roadmap step 35.080 GFLOP; the dominant kernel is the
# c2 forward convolution: 2 * 256*1836 pixels * 32 * (9*32) flop per scene
C2_FLOP_PER_SCENE = 2.0 * 256 * 1836 * 32 * 288
STEP_FLOP_PER_SCENE = 35.080e9            # hidden 128 / latent 64; step_flop_per_scene() for other widths


def step_flop_per_scene():
    """Conv stack fwd+bwd (34.112 GF, no data gradient for c1) + 3 passes x 2 flop x MACs of the four Linear layers."""
    pooled = 32 * 128 * 918 // 4
    return 34.112e9 + 6.0 * (pooled * HIDDEN + HIDDEN * HIDDEN + HIDDEN * LATENT + LATENT * 640000)


def build_model(dev):
    from driving_dirty_amd.autoencoder import BasicAE
    from driving_dirty_amd.roadmap import RoadMapBCE
    torch.manual_seed(SEED)
    ae = BasicAE(Namespace(hidden_dim=HIDDEN, latent_dim=LATENT))
    hp = Namespace(pretrained_ae=ae, unfreeze_epoch_no=0, learning_rate=1e-3, output_img_freq=500, batch_size=BATCH)
    return RoadMapBCE(hp).to(dev)


def synthetic_batch(dev, batch, rank):
    g = torch.Generator(device=dev).manual_seed(SEED + rank)
    views = torch.rand(batch, 6, 3, H, W, generator=g, device=dev)
    road = torch.rand(batch, 800, 800, generator=g, device=dev) < 0.3
    return (tuple(views), tuple({} for _ in range(batch)), tuple(road))


class KernelTimer:
    """HIP events around every launch of the dominant kernel (c2 forward conv) on its launch stream."""

    def __init__(self):
        self.pairs = []
        self.enabled = False

    def install(self):
        from driving_dirty_amd import ops
        # the encoder stack runs c2 forward through conv_wino_fwd_bits (Winograd, default) or conv_fwd_bits (direct)
        def wrap(inner):
            def timed(x, packed, bias, desc):
                hot = self.enabled and desc.cin_real == 32 and desc.stride == 1
                if not hot:
                    return inner(x, packed, bias, desc)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                out = inner(x, packed, bias, desc)
                e.record()
                self.pairs.append((s, e))
                return out
            return timed
        ops.conv_fwd_bits = wrap(ops.conv_fwd_bits)
        ops.conv_wino_fwd_bits = wrap(ops.conv_wino_fwd_bits)
        ops.conv_wino2_fwd_bits = wrap(ops.conv_wino2_fwd_bits)

    def mean_ms(self):
        return sum(s.elapsed_time(e) for s, e in self.pairs) / max(len(self.pairs), 1)


def measured_traffic():
    """HBM bytes per c2-forward launch from the committed rocprofv3 PMC passes (profiles/*_c2_fwd_traffic.json,
    written by tools/pmc_traffic.py); None when no such profile exists.  PMC collection needs the profiler, so it
    cannot be live inside this process."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_c2_fwd_traffic.json")))
    if not files:
        return None
    try:
        return float(json.load(open(files[-1]))["hbm_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        return None


def host_cores():
    """CPU threads this job may really use: affinity, capped by the cgroup quota and by the GPU box's
    per-GPU CPU share (16); DD_CPU_THREADS overrides."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return int(os.environ.get("DD_CPU_THREADS", min(n, 16)))


def cpu_baseline(sample_batch=4, steps=2):
    """The CPU oracle (oracle/: pure-torch restatement of the reference path) timed on this host's cores.

    Bounded sample: the same step (stitch -> encoder -> head -> BCE -> backward -> torch Adam) at bs = 4,
    one warm-up + ``steps`` timed steps (~10-30 s)."""
    from oracle import ae_parts, steps as osteps
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(SEED)
    enc = ae_parts.EncoderNet(HIDDEN, LATENT, 3, H, 6 * W)
    head = torch.nn.Linear(LATENT, 800 * 800)
    opt = torch.optim.Adam(list(enc.parameters()) + list(head.parameters()), lr=1e-3)
    g = torch.Generator().manual_seed(SEED)
    views = torch.rand(sample_batch, 6, 3, H, W, generator=g)
    road = torch.rand(sample_batch, 800, 800, generator=g) < 0.3
    batch = (tuple(views), None, tuple(road))

    def step():
        opt.zero_grad(set_to_none=True)
        loss = osteps.roadmap_bce_loss(enc, head, batch)[0]
        loss.backward()
        opt.step()
    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(sample_batch / dt, 3), "unit": "scenes/s", "cores": cores, "kind": "port",
            "sample": f"oracle roadmap step fwd+bwd+Adam, bs={sample_batch}, {steps} timed steps after 1 warm-up, "
                      f"{dt:.2f} s/step, torch {torch.__version__} CPU"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-adam-overlap", action="store_true", help="run the whole optimizer step after backward")
    ap.add_argument("--cu-budget", type=int, default=0, help="compute units the conv grids may fill (0 = 256, or 240 when N > 1)")
    ap.add_argument("--hidden", type=int, default=HIDDEN, help="encoder hidden width (reference default 128; its other setting: 256)")
    ap.add_argument("--latent", type=int, default=LATENT, help="latent width (64; with --hidden 256: 128)")
    ap.add_argument("--wino-1d", action="store_true", help="c2 forward / data gradient by F(2,3) along x instead of F(2x2,3x3)")
    ap.add_argument("--direct-conv", action="store_true", help="c2 forward / data gradient on the direct kernels instead of Winograd F(2,3)")
    ap.add_argument("--rows-per-task", type=int, default=0, help="conv kernel tuning knob (results unchanged)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("NCCL_MAX_NCHANNELS", os.environ.get("DD_RESERVED_CUS", "16"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # one process per GPU; DD_DIST_BACKEND=gloo + several ranks on ONE card is only a rehearsal of the N > 1
    # control flow on a single-GPU box (the real backend is nccl = RCCL over xGMI)
    backend = os.environ.get("DD_DIST_BACKEND", "nccl")
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE {world}: launch with torch.distributed.run for N > 1"

    from driving_dirty_amd import _lib
    from driving_dirty_amd.ddp import GradSync
    from driving_dirty_amd.optim import HipAdam
    _lib.lib()                                            # fail loudly if the HIP library is missing
    # RCCL's all-reduce workgroups run for milliseconds beside the conv backward and need LDS the conv workgroups do not
    # leave free: GradSync gives them their own compute units (and RCCL is capped at as many channels) from the first big
    # gradient to the end of the step, so the conv grids stay one resident round.  DD_RESERVED_CUS overrides.
    reserve = int(os.environ.get("DD_RESERVED_CUS", "16")) if world > 1 else 0
    if a.cu_budget:
        _lib.check(_lib.lib().dd_set_cu_budget(a.cu_budget), "dd_set_cu_budget")

    globals().update(HIDDEN=a.hidden, LATENT=a.latent)
    if a.direct_conv:
        from driving_dirty_amd import ops as _o
        _o.WINOGRAD = False
    if a.wino_1d:
        from driving_dirty_amd import ops as _o
        _o.WINOGRAD_2D = False
    model = build_model(dev)
    model.ae.encoder.rows_per_task = a.rows_per_task
    model.training_step(synthetic_batch(dev, 2, rank), 0)["loss"].backward()   # unfreezes the AE (epoch 0 >= 0)
    model.zero_grad(set_to_none=True)
    opt = HipAdam(model.parameters(), lr=1e-3)
    sync = GradSync(model, reserve_cus=reserve)
    if not a.no_adam_overlap:
        opt.overlap_with_backward(grad_scale=sync.grad_scale, grad_sync=sync if world > 1 else None)
    batch = synthetic_batch(dev, BATCH, rank)
    timer = KernelTimer()
    timer.install()

    def step(i):
        model.zero_grad(set_to_none=True)
        out = model.training_step(batch, i)
        out["loss"].backward()
        sync.finish()
        opt.step(grad_scale=sync.grad_scale)
        return out["loss"]

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss.detach())

    if rank == 0:
        ms = dt / a.steps * 1e3
        value = world * BATCH * a.steps / dt
        k_ms = timer.mean_ms()
        assert timer.pairs, "the dominant kernel was never launched through the timed entry point"
        achieved = C2_FLOP_PER_SCENE * BATCH / (k_ms * 1e-3) / 1e12      # ALGORITHMIC flops (direct-convolution count)
        from driving_dirty_amd import ops as _ops
        wino = bool(_ops.WINOGRAD)
        wino2 = wino and bool(_ops.WINOGRAD_2D)
        line = {
            "metric": "6-view scenes/sec fwd+bwd, roadmap model bs=32",
            "value": round(value, 2), "unit": "scenes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: roadmap segmentation 6x3x256x306 -> 800x800 mask, "
                                   "bs=32 per GPU, fp32, hidden %d / latent %d, encoder unfrozen, fwd+bwd+Adam" % (HIDDEN, LATENT),
                       "global_batch": world * BATCH, "parallelism": f"dp{world}", "final_loss": round(loss_val, 6)},
            "step_frac_of_fp32_mfma_peak": round(step_flop_per_scene() * BATCH * world / (ms * 1e-3) / 1e12
                                                 / (PEAK_F32_MFMA_TF * world), 4),
            "roofline": {"kernel": ("conv_wino2_fwd (c2 forward, Winograd F(2x2,3x3): issues 4/9 of the algorithmic flops)" if wino2 else
                                    "conv_wino_fwd (c2 forward, Winograd F(2,3) along x: issues 2/3 of the algorithmic flops)"
                                    if wino else "conv_strip_fwd<CIN=32,S=1> (c2 forward, direct)") + ", 74% of encoder FLOPs fwd",
                         "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TF,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TF, 4), "traffic": measured_traffic(),
                         "launch_ms": round(k_ms, 4), "launches_timed": len(timer.pairs),
                         "issued_frac": round(achieved * (4.0 / 9.0 if wino2 else 2.0 / 3.0 if wino else 1.0) / PEAK_F32_MFMA_TF, 4)},
        }
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
