#!/usr/bin/env python3
"""Headline benchmark: 6-view scenes/sec, roadmap model, fwd + bwd + Adam, bs = 32 per GPU, fp32.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4,5}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config C]

``--config`` selects the BASELINE.json entry (default 2 = configs[1], the one the metric is quoted on): 3 = box head on the
frozen encoder (bs 32 per GPU), 4 = joint roadmap + box step (bs 32 per GPU: 256 on 8 GPUs), 5 = bf16 at 2x resolution (bs 16
per GPU: 128 on 8 GPUs).  Every config runs on any N and prints the same JSON shape.

One process per GPU.  Started WITHOUT a launcher (`python bench.py --gpus N`, no WORLD_SIZE in the environment) the
parent process starts its N ranks itself -- N fresh children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before
anything touches a GPU -- waits for them and exits with their status; started by torch.distributed.run it is a rank.
Ranks shard the batch (weak scaling: 32 scenes per GPU); gradients are all-reduced over RCCL
(driving_dirty_amd.ddp.GradSync) overlapped with the conv backward.  A "step" is one pass of the hot path over one
synthetic batch: 6-view stitch -> conv encoder -> pool -> dense blocks -> Linear(64, 640000) -> BCE-with-logits,
backward through everything, Adam on all 162 M parameters.  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line (fields: see the task contract / DESIGN.md section 5).

Workload = BASELINE.json configs[1] (reference src/roadmap_model/roadmap_bce_v2.py with the report's best AE:
hidden 128 / latent 64, FinalReport Table 1), synthetic 6x3x256x306 images and 800x800 road masks, default
PyTorch init under the reference's seed 20200505.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from argparse import Namespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20200505                     # reference autoencoder.py:16-18
BATCH, H, W = 32, 256, 306
HIDDEN, LATENT = 128, 64
PEAK_F32_MFMA_TF = 157.3            # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix)
PEAK_BF16_MFMA_TF = 2500.0          # same guide: dense bf16 matrix peak (config 5 is HBM-bound far below it)
PEAK_HBM_GBS = 8000.0               # same guide: HBM3E peak
# algorithmic work per scene: the 32 -> 32 stride-1 layer, one pass = 2 * 256*1836 pixels * 32 * (9*32) flop;
# the 3 -> 32 layer's weight gradient = 2 * 256*1836 * 32 * 27
C2_FLOP_PER_SCENE = 2.0 * 256 * 1836 * 32 * 288
C1_WGRAD_FLOP_PER_SCENE = 2.0 * 256 * 1836 * 32 * 27
PIXELS_PER_SCENE = 256 * 1836


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5), help="BASELINE.json configs[N-1]: 2 roadmap (the headline), "
                    "3 box head on the frozen encoder, 4 joint roadmap + box, 5 bf16 at 2x resolution")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the config-1 / config-3 / config-5 step timings after the headline")
    ap.add_argument("--adam-overlap", choices=("auto", "on", "off"), default="auto",
                    help="optimizer passes of the big tensors on a side stream beside the backward (on) or after it (off).  auto = TrainStep's "
                    "choice: beside the backward for the fp32 models (config 2, same box, on / off: 7.51-7.62 / 7.86-7.93 ms), after it with four "
                    "workgroups per CU for a bf16 encoder (config 5: same step either way, each kernel at its own roof; "
                    "profiles/r05_config5_adam_overlap_ab.txt)")
    ap.add_argument("--no-adam-overlap", action="store_true", help="= --adam-overlap off")
    ap.add_argument("--adam-blocks-per-cu", type=int, default=0, help="persistent workgroups per CU of the optimizer kernels (0 = TrainStep's "
                    "choice: 1 beside the backward, 4 after it); A/B")
    ap.add_argument("--cu-budget", type=int, default=0, help="compute units the conv grids may fill (0 = 256, or 240 when N > 1)")
    ap.add_argument("--hidden", type=int, default=HIDDEN, help="encoder hidden width (reference default 128; its other setting: 256)")
    ap.add_argument("--latent", type=int, default=LATENT, help="latent width (64; with --hidden 256: 128)")
    ap.add_argument("--wino-1d", action="store_true", help="c2 forward / data gradient by F(2,3) along x instead of F(2x2,3x3)")
    ap.add_argument("--direct-conv", action="store_true", help="c2 forward / data gradient on the direct kernels instead of Winograd")
    ap.add_argument("--rows-per-task", type=int, default=0, help="conv kernel tuning knob (results unchanged)")
    ap.add_argument("--shard-optimizer", choices=("auto", "on", "off"), default="auto",
                    help="N > 1: reduce-scatter + Adam on the owned 1/N + all-gather under the next forward (on) instead of all-reduce + "
                    "replicated Adam (off).  auto = on for the configs with a big gradient message (2, 4, 5: 0.65 / 0.65 / 2.09 GB -- the budget "
                    "of DESIGN.md section 6 favours it at every N and link speed), off for 3 (2 MB of head gradients)")
    ap.add_argument("--factor-linear", choices=("auto", "on", "off"), default="auto",
                    help="data parallel: the big Linear layers all-gather their factors (input, output gradient: 202 MB per rank) instead "
                         "of reducing their weight gradients (648 MB); every rank forms the global-batch gradient itself.  auto = on at "
                         "N = 2 (one xGMI link carries the whole message), where it replaces the sharded optimizer")
    ap.add_argument("--fuse-linear-wgrad", choices=("on", "off"), default="on",
                    help="rank-B optimizer pass: the weight gradients of the big Linear layers are formed inside their Adam pass, never written "
                         "(off: dd_linear_wgrad + dd_adam_step, the round-4 arrangement; A/B)")
    ap.add_argument("--passes-last", choices=("auto", "on", "off"), default="auto",
                    help="c2's data gradient first and the optimizer passes beside its weight gradient, last (auto: with the rank-B pass on one GPU "
                         "or in factor mode); off: the round-4 order; A/B")
    ap.add_argument("--alt-all-reduce", choices=("on", "off"), default="on",
                    help="N > 1 with a sharded / factor-gather default: time the plain all-reduce step first and carry it in the line (on)")
    ap.add_argument("--simulate-shard", type=int, default=0, metavar="N",
                    help="one GPU, timing only: the COMPUTE side of an N-GPU sharded step (Adam on rank 0's 1/N of every big tensor, no "
                    "collectives; the parameters it leaves are meaningless) -- the per-GPU lower bound of the N-GPU step")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ self-launch
def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N rank processes (this process never touches a GPU) and exit
    with their status.  A failed rank is not restarted; the others are terminated."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    deadline = None
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {procs.index(p)} exited with status {code}; stopping the other {len(alive)} rank(s)", file=sys.stderr, flush=True)
                for q in alive:              # a rank died: the collective would hang the others
                    q.terminate()
                deadline = time.monotonic() + 10.0
        if deadline is not None and alive and time.monotonic() > deadline:
            for q in alive:                  # a rank blocked inside a collective or a GPU wait ignores SIGTERM: no GPU holder is left behind
                q.kill()
            for q in alive:
                q.wait()
            alive = []
        time.sleep(0.05)
    return rc


# ------------------------------------------------------------------------------------------------ rank body
def step_flop_per_scene():
    """Conv stack fwd+bwd (34.112 GF, no data gradient for c1) + 3 passes x 2 flop x MACs of the four Linear layers."""
    pooled = 32 * 128 * 918 // 4
    return 34.112e9 + 6.0 * (pooled * HIDDEN + HIDDEN * HIDDEN + HIDDEN * LATENT + LATENT * 640000)


def build_model(dev):
    import torch
    from driving_dirty_amd.autoencoder import BasicAE
    from driving_dirty_amd.roadmap import RoadMapBCE
    torch.manual_seed(SEED)
    ae = BasicAE(Namespace(hidden_dim=HIDDEN, latent_dim=LATENT))
    hp = Namespace(pretrained_ae=ae, unfreeze_epoch_no=0, learning_rate=1e-3, output_img_freq=500, batch_size=BATCH)
    return RoadMapBCE(hp).to(dev)


def synthetic_batch(dev, batch, rank):
    import torch
    g = torch.Generator(device=dev).manual_seed(SEED + rank)
    views = torch.rand(batch, 6, 3, H, W, generator=g, device=dev)
    road = torch.rand(batch, 800, 800, generator=g, device=dev) < 0.3
    return (tuple(views), tuple({} for _ in range(batch)), tuple(road))


class KernelTimer:
    """HIP events around every launch of the two c2 kernels that dominate the step, recorded on their launch stream (the
    current torch stream): the data gradient with the fused c1 weight gradient (the longest kernel of the step) and the forward."""

    def __init__(self):
        self.pairs = {"c2_dgrad_w1": [], "c2_fwd": []}
        self.enabled = False

    def install(self):
        import torch
        from driving_dirty_amd import ops

        def wrap(inner, key, hot):
            def timed(*a, **kw):
                if not (self.enabled and hot(*a)):
                    return inner(*a, **kw)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                out = inner(*a, **kw)
                e.record()
                self.pairs[key].append((s, e))
                return out
            return timed
        is_c2 = lambda x, packed, bias, desc: desc.cin_real == 32 and desc.stride == 1      # noqa: E731
        self._saved = {n: getattr(ops, n) for n in ("conv_fwd_bits", "conv_wino_fwd_bits", "conv_wino2_fwd_bits", "conv_wino2_dgrad_w1")}
        ops.conv_fwd_bits = wrap(ops.conv_fwd_bits, "c2_fwd", is_c2)
        ops.conv_wino_fwd_bits = wrap(ops.conv_wino_fwd_bits, "c2_fwd", is_c2)
        ops.conv_wino2_fwd_bits = wrap(ops.conv_wino2_fwd_bits, "c2_fwd", is_c2)
        ops.conv_wino2_dgrad_w1 = wrap(ops.conv_wino2_dgrad_w1, "c2_dgrad_w1", lambda *a: True)

    def uninstall(self):
        from driving_dirty_amd import ops
        for n, f in getattr(self, "_saved", {}).items():
            setattr(ops, n, f)
        self._saved = {}

    def mean_ms(self, key):
        p = self.pairs[key]
        return sum(s.elapsed_time(e) for s, e in p) / len(p) if p else None


class AbiTimer:
    """HIP events around selected C-ABI calls (``libdd_hotpath.so`` entry points), recorded on the stream the call launches on
    (the current torch stream at the time of the call -- HipAdam's side stream for its early passes).  ``watch`` is
    {key: (entry point, predicate over the ctypes arguments)}; one entry point = one kernel launch for the ones used here."""

    def __init__(self, watch):
        self.watch = watch
        self.pairs = {k: [] for k in watch}
        self.enabled = False

    def install(self):
        import torch
        from driving_dirty_amd import _lib
        lib = _lib.lib()
        by_symbol = {}
        for key, (symbol, pred) in self.watch.items():
            by_symbol.setdefault(symbol, []).append((key, pred))
        self._saved = {}
        for symbol, cases in by_symbol.items():
            inner = getattr(lib, symbol)
            self._saved[symbol] = inner

            def timed(*a, _inner=inner, _cases=cases):
                if self.enabled:
                    for key, pred in _cases:
                        if pred(*a):
                            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            s.record()
                            rc = _inner(*a)
                            e.record()
                            self.pairs[key].append((s, e))
                            return rc
                return _inner(*a)
            setattr(lib, symbol, timed)      # every caller looks the entry point up on this CDLL object

    def uninstall(self):
        from driving_dirty_amd import _lib
        for symbol, inner in getattr(self, "_saved", {}).items():
            setattr(_lib.lib(), symbol, inner)
        self._saved = {}

    def mean_ms(self, key):
        p = self.pairs[key]
        return sum(s.elapsed_time(e) for s, e in p) / len(p) if p else None


def _desc(ref):
    return ref._obj                          # ctypes.byref(struct) -> the struct


def measured_traffic(pattern):
    """HBM bytes per launch from a committed rocprofv3 PMC profile (tools/pmc_traffic.py writes these files; PMC collection
    needs the profiler, so it cannot be live inside this process) -> (bytes, "profiles/<file>") or (None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None, None
    try:
        return float(json.load(open(files[-1]))["hbm_bytes_per_launch"]), "profiles/" + os.path.basename(files[-1])
    except (OSError, ValueError, KeyError):
        return None, None


def live_traffic(case, kernel_pattern, timeout_s=60):
    """HBM bytes per launch of one kernel, measured NOW on this box: two child runs of `rocprofv3 --kernel-trace --pmc <counter>
    -- python3 tools/bench_one.py <case>` (FETCH_SIZE, WRITE_SIZE: separate passes, MI355X_MICROARCH.md HBM section; FETCH_SIZE
    doubled: gfx950 tallies 128-byte requests as 64).  The children are ordinary child processes (no exec from this process).
    -> (bytes, description) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    if os.environ.get("DD_BENCH_LIVE_TRAFFIC", "1") == "0":
        return None, "disabled (DD_BENCH_LIVE_TRAFFIC=0)"
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None:
        return None, "rocprofv3 not found"
    med = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="dd_pmc_", dir="/tmp")
            cmd = [prof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                   os.path.join(ROOT, "tools", "bench_one.py"), case]
            # its own process group: on a timeout the profiler AND the program it started are killed (no GPU holder left behind)
            child = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                     start_new_session=True)
            try:
                rc = child.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(child.pid, signal.SIGKILL)
                child.wait()
                return None, f"rocprofv3 --pmc {counter} pass killed after {timeout_s} s"
            files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
            if rc != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {rc})"
            per = {}
            for row in csv.DictReader(open(files[0])):
                if kernel_pattern in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            shutil.rmtree(d, ignore_errors=True)
            if not per:
                return None, f"no launch of {kernel_pattern} in the {counter} pass"
            vals = sorted(per.values())
            med[counter] = vals[len(vals) // 2] * 1024.0
    except (OSError, subprocess.SubprocessError, KeyError, ValueError) as e:
        return None, f"live PMC pass failed: {type(e).__name__}"
    return 2.0 * med["FETCH_SIZE"] + med["WRITE_SIZE"], (f"live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate child passes in this run) over "
                                                         f"tools/bench_one.py {case}; FETCH_SIZE x 2 (gfx950), median over launches")


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


def cpu_baseline(sample_batch=BATCH, steps=1, ae_batch=4, ae_steps=3):
    """The CPU oracle (oracle/: pure-torch restatement of the reference path) timed on this host's cores: the headline config-2 step AT
    THE METRIC'S BATCH (bs = 32: the same step the GPU line measures; 1 warm-up + 1 timed step -- 49 s per step on the 8
    cores of the build container, 13.5 GB resident) and the config-1 autoencoder step at ITS batch (bs = 4: BASELINE.json configs[0]),
    1 warm-up + 3 timed."""
    import numpy as np
    import torch
    from oracle import ae_parts, steps as osteps
    cores = host_cores()
    torch.set_num_threads(cores)

    def timed(step, n):
        step()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        return (time.perf_counter() - t0) / n

    torch.manual_seed(SEED)
    enc = ae_parts.EncoderNet(HIDDEN, LATENT, 3, H, 6 * W)
    head = torch.nn.Linear(LATENT, 800 * 800)
    opt = torch.optim.Adam(list(enc.parameters()) + list(head.parameters()), lr=1e-3)
    g = torch.Generator().manual_seed(SEED)
    views = torch.rand(sample_batch, 6, 3, H, W, generator=g)
    road = torch.rand(sample_batch, 800, 800, generator=g) < 0.3
    batch = (tuple(views), None, tuple(road))

    def roadmap_step():
        opt.zero_grad(set_to_none=True)
        osteps.roadmap_bce_loss(enc, head, batch)[0].backward()
        opt.step()
    dt = timed(roadmap_step, steps)
    out = {"value": round(sample_batch / dt, 3), "unit": "scenes/s", "cores": cores, "kind": "port",
           "port_of": "oracle/ (plain-torch CPU restatement of the reference path; the reference itself never travels to the GPU box): "
                      "held to the reference's own outputs by tests/golden/*.npz, which tests/golden/make_golden.py generates by "
                      "importing /root/reference (tests/test_oracle_golden.py: fp32 2e-6, fp64 1e-12)",
           "sample": f"oracle roadmap step (config 2) fwd+bwd+Adam, bs={sample_batch}, {steps} timed steps after 1 warm-up, "
                     f"{dt:.2f} s/step, torch {torch.__version__} CPU"}
    del opt, head, batch, road
    views = views[:ae_batch].clone()
    # config 1 (BASELINE.json configs[0]: the reference's own CPU-runnable case): masked-view autoencoder step, bs = 4
    dec = ae_parts.DecoderNet(HIDDEN, LATENT, 3, H, W)
    opt = torch.optim.Adam(list(enc.parameters()) + list(dec.parameters()), lr=1e-3)
    rng = np.random.RandomState(SEED)

    def ae_step():
        opt.zero_grad(set_to_none=True)
        osteps.ae_loss(enc, dec, views, rng)[0].backward()
        opt.step()
    dt1 = timed(ae_step, ae_steps)
    out["config1_ae"] = {"value": round(ae_batch / dt1, 3), "unit": "scenes/s", "cores": cores, "kind": "port",
                         "sample": f"oracle BasicAE step (config 1: src/autoencoder/autoencoder.py, bs={ae_batch}) fwd+bwd+Adam, "
                                   f"{ae_steps} timed steps after 1 warm-up, {dt1:.2f} s/step"}
    return out


def time_steps(step, steps, warmup):
    import torch
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


_T0 = time.perf_counter()


def _phase(name):
    if os.environ.get("DD_BENCH_PHASES") == "1":
        sys.stderr.write(f"bench.py phase {name}: {time.perf_counter() - _T0:.1f} s since start\n")
        sys.stderr.flush()


def other_configs(a, dev, steps=10, warmup=3):
    """Per-GPU step rates of the other BASELINE configurations, measured in this process after the headline timing (same
    synthetic-data conventions, the same TrainStep; diagnostic numbers, not `value`).  Configs 3 / 4 / 5 carry the `roofline` of their
    dominant kernel, from HIP events around its C-ABI launches inside their timed steps (AbiTimer)."""
    import torch
    from driving_dirty_amd.autoencoder import BasicAE
    from driving_dirty_amd.roadmap import RoadMapBCE
    from driving_dirty_amd.train import TrainStep
    res = {}

    def run(name, model, batch, nbatch, extra=None, watch=None, cfg=None):
        ts = TrainStep(model, lr=1e-3, scheduler=False, fuse_linear_wgrad=getattr(a, "fuse_linear_wgrad", "on") == "on")
        timer = AbiTimer(watch) if watch else None
        if timer:
            timer.install()

        def step(i):
            if timer:
                timer.enabled = i >= warmup
            ts(batch, i)
        dt = time_steps(step, steps, warmup)
        entry = dict({"ms_per_step": round(dt * 1e3, 3), "scenes_per_s": round(nbatch / dt, 1), "batch": nbatch, "steps": steps, "warmup": warmup},
                     **(extra or {}))
        if timer:
            timer.enabled = False
            entry["roofline"] = watched_roofline(cfg, timer, nbatch, dt * 1e3)
            timer.uninstall()
        res[name] = entry
        ts.close()
        _phase(name)

    # config 1's GPU twin: BasicAE masked-view pre-training step (autoencoder.py:78-93), fwd+bwd+Adam
    for b in (4, BATCH):
        torch.manual_seed(SEED)
        ae = BasicAE(Namespace(hidden_dim=HIDDEN, latent_dim=LATENT, learning_rate=1e-3, output_img_freq=500)).to(dev)
        run(f"config1_ae_pretrain_bs{b}", ae, torch.rand(b, 6, 3, H, W, device=dev), b, {"dtype": "f32"})
        del ae
        torch.cuda.empty_cache()
    names = {3: "config3_bbox_frozen_encoder_bs32", 4: "config4_joint_roadmap_bbox_bs32_per_gpu", 5: "config5_bf16_2x_resolution_bs16"}
    for c in (3, 4, 5):
        cfg = setup_config(Namespace(config=c, rows_per_task=0), dev, 0)
        extra = {"dtype": "bf16 (fp32 master weights, accumulate, tail)" if c == 5 else "f32"}
        algo = cfg["flop_per_scene"] * cfg["per_gpu_batch"]
        peak = PEAK_BF16_MFMA_TF if c == 5 else PEAK_F32_MFMA_TF
        run(names[c], cfg["model"], cfg["batch"], cfg["per_gpu_batch"], extra, cfg["watch"], c)
        ms = res[names[c]]["ms_per_step"]
        res[names[c]].update(algorithmic_TFLOPs=round(algo / (ms * 1e-3) / 1e12, 1),
                             **{("frac_bf16_mfma_peak" if c == 5 else "frac_fp32_mfma_peak"): round(algo / (ms * 1e-3) / 1e12 / peak, 4)})
        if c in (3, 4):
            # precision mode "fp32x3" (csrc/dconv_split.hip; never the headline): the same step with up_conv_1 / up_conv_2 taking every fp32
            # product as six bf16 x bf16 products (exact 3-way operand split, fp32 accumulate) on the bf16 matrix pipe
            watch = {"up_conv_1_fwd_split": ("dd_dconv_fwd_split", lambda *x: _desc(x[6]).cin == 96 and _desc(x[6]).pad_h > 0),
                     "up_conv_2_fwd_split": ("dd_dconv_fwd_split", lambda *x: _desc(x[6]).cin == 64 and _desc(x[6]).cout == 32 and _desc(x[6]).pad_h > 0),
                     "up_conv_1_dgrad_split": ("dd_dconv_fwd_split", lambda *x: _desc(x[6]).cin == 64 and _desc(x[6]).cout == 96 and _desc(x[6]).pad_h == 0),
                     "up_conv_2_dgrad_split": ("dd_dconv_fwd_split", lambda *x: _desc(x[6]).cin == 32 and _desc(x[6]).cout == 64 and _desc(x[6]).pad_h == 0),
                     "up_conv_3_dgrad_split": ("dd_dconv_fwd_split", lambda *x: _desc(x[6]).cin == 16 and _desc(x[6]).cout == 32 and _desc(x[6]).pad_h == 0),
                     "up_conv_1_wgrad_split": ("dd_dconv_wgrad_split", lambda *x: x[6] == 96),
                     "up_conv_2_wgrad_split": ("dd_dconv_wgrad_split", lambda *x: x[6] == 64),
                     "split_input_pass": ("dd_dconv_split_input", lambda *x: True), "split_rows_pass": ("dd_dconv_split_rows", lambda *x: True)}
            cfg["model"].box_merge.precision = "fp32x3"      # the documented mode switch (hparams.precision = "fp32x3" sets the same attribute)
            try:
                run("config3_bbox_split_products_bs32" if c == 3 else "config4_joint_split_products_bs32_per_gpu", cfg["model"], cfg["batch"], cfg["per_gpu_batch"],
                    {"dtype": "f32 (bf16x6 split products, fp32 accumulate) in the forward, data gradient and weight gradient of up_conv_1 and "
                              "up_conv_2; everything else exact fp32"}, watch, "3s")
            finally:
                cfg["model"].box_merge.precision = "fp32"
        del cfg
        torch.cuda.empty_cache()
    # config 2 at the reference's DEFAULT width (autoencoder.py:33-34,164-166: hidden 256 / latent 128; SURVEY.md 8d "also report 256/128"):
    # fc1 962 MB, head 328 MB, the encoder tail on the separate Linear / BatchNorm kernels
    torch.manual_seed(SEED)
    ae = BasicAE(Namespace(hidden_dim=256, latent_dim=128))
    m = RoadMapBCE(Namespace(pretrained_ae=ae, unfreeze_epoch_no=0, learning_rate=1e-3, output_img_freq=500)).to(dev)
    run("config2_hidden256_latent128_bs32", m, synthetic_batch(dev, BATCH, 0), BATCH, {"dtype": "f32"})
    del m, ae
    torch.cuda.empty_cache()
    res["config2_u8_h2d"] = u8_h2d_step(dev, steps, warmup)
    _phase("config2_u8_h2d")
    return res


def u8_h2d_step(dev, steps, warmup):
    """PCIe-INCLUSIVE rate of the headline step (never `value`): every batch starts in pinned HOST memory as decoded uint8 camera
    frames (the collate's tuple of 32 x [6,256,306,3]) + bool road masks, crosses PCIe on a copy stream one batch ahead of the step
    (driving_dirty_amd.prefetch.DevicePrefetcher) and is read by the model as it is (ToTensor's /255 fused into the 6-view gather)."""
    import torch
    from driving_dirty_amd.prefetch import DevicePrefetcher
    from driving_dirty_amd.train import TrainStep
    model = build_model(dev)
    ts = TrainStep(model, lr=1e-3, scheduler=False)
    g = torch.Generator().manual_seed(SEED)
    host_tuple, host_stacked = [], []
    for _ in range(2):
        frames = torch.randint(0, 256, (BATCH, 6, H, W, 3), dtype=torch.uint8, generator=g)
        road = torch.rand(BATCH, 800, 800, generator=g) < 0.3
        # (a) the reference's collate as it is: a tuple of per-sample tensors, each pinned by itself (64 small copies per batch)
        host_tuple.append((tuple(f.clone().pin_memory() for f in frames), tuple({} for _ in range(BATCH)), tuple(r.clone().pin_memory() for r in road)))
        # (b) a collate that stacks: one pinned [B,6,H,W,3] block + one [B,800,800] block (2 copies per batch); training_step takes both forms
        host_stacked.append((frames.pin_memory(), tuple({} for _ in range(BATCH)), road.pin_memory()))
    nbytes = BATCH * (6 * H * W * 3 + 800 * 800)

    def timed(host, unstack_road):
        t0 = None
        for i, batch in enumerate(DevicePrefetcher((host[i & 1] for i in range(warmup + steps)), dev)):
            if i == warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            if unstack_road:
                batch = (batch[0], batch[1], tuple(batch[2]))      # views of the one device block: the loss reads them through its pointer table
            ts(batch, i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps
    dt_tuple = timed(host_tuple, False)
    dt = timed(host_stacked, True)
    ts.close()
    return {"ms_per_step": round(dt * 1e3, 3), "scenes_per_s": round(BATCH / dt, 1), "batch": BATCH, "steps": steps, "warmup": warmup, "dtype": "f32",
            "input": "uint8 frames [B,6,256,306,3] + bool road masks [B,800,800], each ONE pinned host block (a stacking collate), prefetched one batch "
                     "ahead on a copy stream: 2 copies per batch",
            "per_sample_copies": {"ms_per_step": round(dt_tuple * 1e3, 3), "scenes_per_s": round(BATCH / dt_tuple, 1),
                                  "input": "the reference's collate as it is: tuples of per-sample pinned tensors, 64 copies per batch"},
            "pcie_bytes_per_step": nbytes, "pcie_GBs_needed": round(nbytes / dt / 1e9, 2)}


UPCONV1_FLOP_PER_SCENE = 2.0 * 256 * 256 * 49 * 96 * 64      # RoadMapBoxesMergingCNN.up_conv_1 (components.py:135), one pass
UPCONV2_FLOP_PER_SCENE = 2.0 * 298 * 298 * 49 * 64 * 32      # up_conv_2 (components.py:136): input 298 x 298


def box_batch(dev, batch, rank):
    """Config 3 / 4 inputs: views + road masks as in config 2, box targets pre-rasterised (the rasteriser is
    outside the timed region)."""
    import torch
    sample, _, road = synthetic_batch(dev, batch, rank)
    g = torch.Generator(device=dev).manual_seed(SEED + 1000 + rank)
    tgt = tuple({"bb_map": (torch.rand(800, 800, generator=g, device=dev) < 0.02).float()} for _ in range(batch))
    return (sample, tgt, road)


def setup_config(a, dev, rank):
    """-> dict(model, batch, per_gpu_batch, metric, workload, dtype, flop_per_scene, timers): the BASELINE.json entry ``a.config``."""
    import torch
    from driving_dirty_amd.autoencoder import BasicAE
    cfg = a.config
    if cfg == 2:
        model = build_model(dev)
        model.ae.encoder.rows_per_task = a.rows_per_task
        return dict(model=model, batch=synthetic_batch(dev, BATCH, rank), per_gpu_batch=BATCH, dtype="f32",
                    metric="6-view scenes/sec fwd+bwd, roadmap model bs=32",
                    workload="BASELINE configs[1]: roadmap segmentation 6x3x256x306 -> 800x800 mask, bs=32 per GPU, fp32, "
                             "hidden %d / latent %d, encoder unfrozen, fwd+bwd+Adam" % (HIDDEN, LATENT),
                    flop_per_scene=step_flop_per_scene())
    torch.manual_seed(SEED)
    if cfg in (3, 4):
        from driving_dirty_amd.joint import JointRoadMapBBox
        from driving_dirty_amd.spatial import BBSpatialRoadMap
        ae = BasicAE(Namespace(hidden_dim=HIDDEN, latent_dim=LATENT))
        if cfg == 3:
            model = BBSpatialRoadMap(Namespace(pretrained_ae=ae, unfreeze_epoch_no=10 ** 9, learning_rate=1e-3, output_img_freq=500,
                                               mse_loss=False)).to(dev)
            flop = (69.14 + 138.28 + 11.64) * 1e9          # head fwd + head bwd + frozen encoder fwd
            what = ("BASELINE configs[2]: bounding-box head (SpatialMappingCNN + RoadMapBoxesMergingCNN) on the frozen AE encoder, "
                    "6-view input, bs=32 per GPU, fp32, fwd+bwd+Adam on the heads")
            metric = "6-view scenes/sec fwd+bwd, bbox head on frozen encoder bs=32"
        else:
            model = JointRoadMapBBox(Namespace(pretrained_ae=ae, learning_rate=1e-3, output_img_freq=500)).to(dev)
            flop = step_flop_per_scene() + (69.14 + 138.28) * 1e9      # roadmap step + head fwd + head bwd (ss_conv's 0.8 GF data gradient into the encoder not counted)
            what = ("BASELINE configs[3]: joint roadmap + bounding-box multi-task step (one shared encoder pass, both heads, summed "
                    "losses), bs=32 per GPU (256 on 8 GPUs), fp32, fwd+bwd+Adam")
            metric = "6-view scenes/sec fwd+bwd, joint roadmap+bbox bs=32 per GPU"
        watch = {"up_conv_1_fwd": ("dd_dconv_fwd", lambda *x: _desc(x[5]).cin == 96 and _desc(x[5]).cout == 64),
                 "up_conv_1_dgrad": ("dd_dconv_fwd", lambda *x: _desc(x[5]).cin == 64 and _desc(x[5]).cout == 96),
                 "up_conv_1_wgrad": ("dd_dconv_wgrad", lambda *x: x[8] == 96 and x[13] == 64),
                 "up_conv_2_fwd": ("dd_dconv_fwd", lambda *x: _desc(x[5]).cin == 64 and _desc(x[5]).cout == 32 and _desc(x[5]).kh == 7),
                 "up_conv_2_dgrad": ("dd_dconv_fwd", lambda *x: _desc(x[5]).cin == 32 and _desc(x[5]).cout == 64 and _desc(x[5]).kh == 7),
                 "up_conv_2_wgrad": ("dd_dconv_wgrad", lambda *x: x[8] == 64 and x[13] == 32 and x[14] == 7)}
        return dict(model=model, batch=box_batch(dev, BATCH, rank), per_gpu_batch=BATCH, dtype="f32", metric=metric, workload=what,
                    flop_per_scene=flop, watch=watch)
    # config 5: bf16 mixed precision at 2x resolution (6x3x512x612), bs = 16 per GPU (128 on 8 GPUs), roadmap step
    from driving_dirty_amd.roadmap import RoadMapBCE
    h2, w2, b5 = 2 * H, 2 * W, 16
    ae = BasicAE(Namespace(hidden_dim=HIDDEN, latent_dim=LATENT, input_height=h2, input_width=6 * w2, output_height=h2, output_width=w2))
    model = RoadMapBCE(Namespace(pretrained_ae=ae, precision="bf16", unfreeze_epoch_no=0, learning_rate=1e-3, output_img_freq=10 ** 9)).to(dev)
    g = torch.Generator(device=dev).manual_seed(SEED + rank)
    batch = (tuple(torch.rand(b5, 6, 3, h2, w2, generator=g, device=dev)), None, tuple(torch.rand(b5, 800, 800, generator=g, device=dev) < 0.3))
    pooled2 = 32 * 256 * 1836 // 4
    flop = 136.448e9 + 6.0 * (pooled2 * HIDDEN + HIDDEN * HIDDEN + HIDDEN * LATENT + LATENT * 640000)
    is_c2 = lambda d: d.cin_real == 32 and d.stride == 1      # noqa: E731
    watch = {"adam_fc1": ("dd_adam_step", lambda *x: x[4] >= 100_000_000),
             "adam_fc1_rankb": ("dd_adam_step_rankb", lambda *x: x[6] * x[7] >= 100_000_000),
             "c2_fwd_bf16": ("dd_conv_bf16_fwd", lambda *x: is_c2(_desc(x[5]))),
             "c2_dgrad_bf16": ("dd_conv_bf16_dgrad", lambda *x: is_c2(_desc(x[4]))),
             "c2_wgrad_bf16": ("dd_conv_bf16_wgrad", lambda *x: is_c2(_desc(x[4])))}
    return dict(model=model, batch=batch, per_gpu_batch=b5, dtype="bf16", metric="6-view scenes/sec fwd+bwd, roadmap model bf16 2x resolution bs=16 per GPU",
                workload="BASELINE configs[4]: bf16 mixed precision (fp32 master weights, accumulation, FC tail), 2x input resolution "
                         "6x3x512x612, bs=16 per GPU (128 on 8 GPUs), roadmap step fwd+bwd+Adam", flop_per_scene=flop, watch=watch)


def config2_roofline(timer, ops_mod):
    """The two c2 kernels of the headline step -> (dominant-kernel roofline dict)."""
    wino = bool(ops_mod.WINOGRAD)
    wino2 = wino and bool(ops_mod.WINOGRAD_2D)
    issue = 4.0 / 9.0 if wino2 else 2.0 / 3.0 if wino else 1.0      # share of the direct form's multiplies a Winograd kernel issues
    fwd_ms, dg_ms = timer.mean_ms("c2_fwd"), timer.mean_ms("c2_dgrad_w1")
    assert fwd_ms is not None, "the c2 forward kernel was never launched through the timed entry point"
    kernels = {}
    # the kernels behind the two entry points (csrc/conv3x3.hip launch_wino2): register-row form for the forward and the fused
    # data gradient; DD_WINO2_RING / DD_WINO2_RING_W1 switch to the ring form for A/B
    k_fwd = "conv_wino2_fwd" if os.environ.get("DD_WINO2_RING") else "conv_wino2r_fwd"
    k_w1 = "conv_wino2_fwd" if (os.environ.get("DD_WINO2_RING") or os.environ.get("DD_WINO2_RING_W1")) else "conv_wino2r_fwd"
    algo = C2_FLOP_PER_SCENE * BATCH
    fwd_bytes = PIXELS_PER_SCENE * BATCH * 260.0       # reads a1 (128 B/pixel), writes a2 (128 B/pixel) + one sign word per pixel
    kernels["c2_forward"] = {
        "kernel": k_fwd + "<BIAS_RELU_BITS>" if wino2 else "conv_wino_fwd" if wino else "conv_strip_fwd<32,1>",
        "launch_ms": round(fwd_ms, 4), "launches_timed": len(timer.pairs["c2_fwd"]),
        "issued_TFLOPs": round(algo * issue / (fwd_ms * 1e-3) / 1e12, 2), "algorithmic_equiv_TFLOPs": round(algo / (fwd_ms * 1e-3) / 1e12, 2),
        "mfma_frac": round(algo * issue / (fwd_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TF, 4),
        "hbm_frac_algorithmic": round(fwd_bytes / (fwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
    if dg_ms is not None:
        # c2's data gradient by F(2x2,3x3) (4/9 of the direct multiplies) + c1's weight gradient taken from the masked outputs in
        # registers: 64 more MFMAs on every 256.  Reads g2 (128 B/pixel), one sign word and 16 B of image per pixel.
        issued = algo * issue * (320.0 / 256.0)
        kernels["c2_dgrad_w1"] = {
            "kernel": k_w1 + "<RELU_BITS_W1> (c2 data gradient + fused c1 weight gradient)",
            "launch_ms": round(dg_ms, 4), "launches_timed": len(timer.pairs["c2_dgrad_w1"]),
            "issued_TFLOPs": round(issued / (dg_ms * 1e-3) / 1e12, 2),
            "algorithmic_equiv_TFLOPs": round((algo + C1_WGRAD_FLOP_PER_SCENE * BATCH) / (dg_ms * 1e-3) / 1e12, 2),
            "mfma_frac": round(issued / (dg_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TF, 4),
            "hbm_frac_algorithmic": round(PIXELS_PER_SCENE * BATCH * 148.0 / (dg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
    dom = max(kernels, key=lambda k: kernels[k]["launch_ms"])          # the kernel with the largest in-step time
    traffic, source = measured_traffic("*_c2_dgrad_w1_traffic.json" if dom == "c2_dgrad_w1" else "*_c2_fwd_traffic.json")      # committed profile
    k = kernels[dom]
    k["_pmc"] = ("wino2_dgrad_w1", k_w1 + "<9, 4") if dom == "c2_dgrad_w1" else ("wino2_fwd", k_fwd + "<5, 4")
    # hbm_frac: the MEASURED HBM bytes per launch (rocprofv3 PMC, committed profile) over this run's launch time when a traffic
    # file exists, the algorithmic bytes otherwise
    hbm_frac = round(traffic / (k["launch_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if traffic else k["hbm_frac_algorithmic"]
    # achieved = MFMA flops the kernel ISSUES per launch (a Winograd kernel issues 4/9 of the direct form's) / its mean launch
    # duration from HIP events inside the timed region: frac <= 1 is the share of the fp32 matrix pipe in use.
    # `algorithmic_equiv` is the direct-convolution flop count over the same time.
    pmc = k.pop("_pmc")
    roof = {"kernel": k["kernel"], "bound": "mfma", "achieved": k["issued_TFLOPs"], "peak": PEAK_F32_MFMA_TF, "unit": "TFLOP/s",
            "frac": k["mfma_frac"], "algorithmic_equiv": k["algorithmic_equiv_TFLOPs"], "hbm_frac": hbm_frac,
            "hbm_frac_algorithmic": k["hbm_frac_algorithmic"], "launch_ms": k["launch_ms"], "launches_timed": k["launches_timed"],
            "traffic": traffic, "traffic_source": source, "kernels": kernels}
    return roof, pmc


def refresh_traffic(roof, pmc):
    """Replace the committed-profile traffic figure by one measured in this run (after every model of this process is gone)."""
    live, how = live_traffic(*pmc)
    if live is not None:
        roof["traffic_committed_profile"] = roof["traffic"]
        roof["traffic"], roof["traffic_source"] = live, how
        roof["hbm_frac"] = round(live / (roof["launch_ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
    else:
        roof["traffic_live"] = how


def config5_step_bytes(per_gpu_batch, shard_world=1, rankb=True):
    """Algorithmic HBM bytes of one config-5 step (bf16 activations, fp32 parameters): every activation
    written once and read once per consumer, the two big weights read twice and their gradients written once, Adam's seven
    passes over the parameters it owns.  P = full-resolution pixels of the batch (512 x 3672 per scene), q = P / 4."""
    P = 512 * 3672 * per_gpu_batch
    per_pixel = (20      # stitch: 12 B of fp32 views in, 8 B of bf16 NHWC4 out
                 + 76    # c1 forward: 8 in, 64 out, 4 of sign bits
                 + 132   # c2 forward: 64 in, 64 out, 4 of sign bits
                 + 81    # c3 forward (stride 2): 64 in at P, 64 + 4 out at q
                 + 25    # pool forward: 64 in, 32 (fp32 pooled) + 4 (routing codes) out, at q
                 + 25    # pool backward
                 + 84    # c3 data gradient: 64 in at q, sign bits 4 and 64 out at P
                 + 80    # c3 weight gradient: 64 at P + 64 at q
                 + 132   # c2 data gradient
                 + 128   # c2 weight gradient
                 + 72)   # c1 weight gradient: 8 + 64
    fc1 = (32 * 256 * 1836 // 4) * HIDDEN * 4.0
    head = LATENT * 640000 * 4.0
    if rankb and shard_world == 1:      # rank-B optimizer pass: no dW written or read back (weights read twice; p, m, v read + written)
        return P * float(per_pixel) + 2.0 * (fc1 + head) + 6.0 * (fc1 + head)
    return P * float(per_pixel) + 3.0 * (fc1 + head) + 7.0 * (fc1 + head) / shard_world


def watched_roofline(cfg, timer, per_gpu_batch, step_ms=None):
    """Configs 3 / 4 / 5: the watched C-ABI launches -> the dominant kernel's roofline.  Config 5 (bf16) prices every kernel against
    BOTH roofs -- the time its algorithmic flops need at the dense bf16 matrix peak and its algorithmic bytes at the HBM peak -- and
    reports the one that binds (the larger): min(MFMA, HBM) as SURVEY.md 8d asks."""
    kernels = {}
    for key in timer.pairs:
        ms = timer.mean_ms(key)
        if ms is None:
            continue
        entry = {"launch_ms": round(ms, 4), "launches_timed": len(timer.pairs[key])}
        if cfg == "3s":
            if key in ("split_input_pass", "split_rows_pass"):
                entry.update(kernel="split_input_kernel (fp32 -> three bf16 planes; all of the step's launches averaged)", bound="hbm")
            else:
                flop = {"up_conv_1": UPCONV1_FLOP_PER_SCENE, "up_conv_2": UPCONV2_FLOP_PER_SCENE, "up_conv_3": 2.0 * 340 * 340 * 49 * 32 * 16}[key[:9]] * per_gpu_batch
                # 6 bf16 products per fp32 product: the matrix work ISSUED is 6x the algorithmic flops, priced at the dense bf16 peak
                kname = "dconv_sgfwd_kernel" if "dgrad" in key else "dconv_swgrad_kernel" if "wgrad" in key else "dconv_stfwd_kernel"
                what = "data gradient" if "dgrad" in key else "weight gradient" if "wgrad" in key else "forward"
                entry.update(kernel="%s (%s %s, split products)" % (kname, key[:9], what), bound="mfma", achieved=round(6 * flop / (ms * 1e-3) / 1e12, 1),
                             peak=PEAK_BF16_MFMA_TF, unit="TFLOP/s (bf16 products issued)", frac=round(6 * flop / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TF, 4),
                             fp32_equivalent_TFLOPs=round(flop / (ms * 1e-3) / 1e12, 1))
        elif cfg in (3, 4):
            layer, what = key.rsplit("_", 1)
            flop = {"up_conv_1": UPCONV1_FLOP_PER_SCENE, "up_conv_2": UPCONV2_FLOP_PER_SCENE}[layer] * per_gpu_batch
            entry.update(kernel={"up_conv_1_fwd": "dconv_tfwd_kernel (up_conv_1 forward, input-aligned)",
                                 "up_conv_1_dgrad": "dconv_gfwd_kernel (up_conv_1 data gradient)",
                                 "up_conv_1_wgrad": "dconv_wgrad_kernel (up_conv_1 weight gradient)",
                                 "up_conv_2_fwd": "dconv_tfwd_kernel (up_conv_2 forward)",
                                 "up_conv_2_dgrad": "dconv_mwin_kernel (up_conv_2 data gradient, gather form, 3 phase rows per task)",
                                 "up_conv_2_wgrad": "dconv_wgrad_kernel (up_conv_2 weight gradient)"}[key],
                         bound="mfma", achieved=round(flop / (ms * 1e-3) / 1e12, 2), peak=PEAK_F32_MFMA_TF, unit="TFLOP/s",
                         frac=round(flop / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TF, 4))
        else:
            px = 512 * 3672 * per_gpu_batch
            pooled2 = 32 * 256 * 1836 // 4
            nbytes = {"adam_fc1": 28.0 * pooled2 * HIDDEN,      # p, g, m, v read; p, m, v written: 7 x 4 B per element
                      "adam_fc1_rankb": 24.0 * pooled2 * HIDDEN + 4.0 * per_gpu_batch * (pooled2 + HIDDEN),      # p, m, v read + written; the factors read
                      "c2_fwd_bf16": px * (64 + 64 + 4.0),                    # bf16 NHWC in + out, one sign word per pixel
                      "c2_dgrad_bf16": px * (64 + 64 + 4.0),
                      "c2_wgrad_bf16": px * (64 + 64.0)}[key]
            flop = 0.0 if key.startswith("adam_fc1") else 2.0 * px * 32 * 288
            t_hbm, t_mfma = nbytes / (PEAK_HBM_GBS * 1e9), flop / (PEAK_BF16_MFMA_TF * 1e12)
            hbm_frac = round(nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
            mfma_frac = round(flop / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TF, 4)
            name = {"adam_fc1": "adam_kernel (fc1.fc1.weight, 481 M elements, side stream)",
                    "adam_fc1_rankb": "adam_rankb_lds_kernel (fc1.fc1.weight, 481 M elements: gradient formed in the pass, side stream)", "c2_fwd_bf16": "conv_bf16_fwd (c2)",
                    "c2_dgrad_bf16": "conv_bf16_dgrad (c2)", "c2_wgrad_bf16": "conv_bf16_wgrad (c2)"}[key]
            if t_hbm >= t_mfma:
                entry.update(kernel=name, bound="hbm", achieved=round(nbytes / (ms * 1e-3) / 1e9, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=hbm_frac)
            else:
                entry.update(kernel=name, bound="mfma", achieved=round(flop / (ms * 1e-3) / 1e12, 1), peak=PEAK_BF16_MFMA_TF, unit="TFLOP/s", frac=mfma_frac)
            entry.update(hbm_frac=hbm_frac, mfma_frac=mfma_frac, roof_ms={"hbm": round(t_hbm * 1e3, 4), "mfma": round(t_mfma * 1e3, 4)})
        kernels[key] = entry
    if not kernels:
        return None
    dom = max((k for k in kernels if "achieved" in kernels[k]), key=lambda k: kernels[k]["launch_ms"])
    k = kernels[dom]
    roof = {"kernel": k["kernel"], "bound": k["bound"], "achieved": k["achieved"], "peak": k["peak"], "unit": k["unit"], "frac": k["frac"],
            "launch_ms": k["launch_ms"], "launches_timed": k["launches_timed"], "traffic": None, "kernels": kernels}
    if cfg == 5 and step_ms:
        rankb = "adam_fc1_rankb" in kernels
        nbytes = config5_step_bytes(per_gpu_batch, rankb=rankb)
        roof["step"] = {"algorithmic_bytes": nbytes, "TBps": round(nbytes / (step_ms * 1e-3) / 1e12, 3),
                        "hbm_frac": round(nbytes / (step_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                        "of_which_adam_bytes": (6.0 if rankb else 7.0) * ((32 * 256 * 1836 // 4) * HIDDEN + LATENT * 640000) * 4.0}
    return roof


class RankStderr:
    """Prefix every line this rank writes to stderr (tracebacks included) with its rank, so the output of N processes that
    share one terminal or log can be told apart."""

    def __init__(self, rank, raw):
        self.tag, self.raw, self.bol = f"[rank {rank}] ", raw, True

    def write(self, text):
        out = []
        for ch in text.splitlines(keepends=True):
            out.append((self.tag if self.bol else "") + ch)
            self.bol = ch.endswith("\n")
        self.raw.write("".join(out))
        return len(text)

    def flush(self):
        self.raw.flush()

    def __getattr__(self, name):
        return getattr(self.raw, name)


class Watchdog:
    """A rank that makes no progress for ``limit`` seconds (a peer that never joined the rendezvous, a collective whose partner died,
    a kernel that never ends) prints where it stands and EXITS with status 3 -- `os._exit`: never an exec, the process has touched
    the GPU -- so that the launcher (or torch.distributed.run) tears the job down instead of sitting out the driver's limit in
    silence.  ``beat(phase)`` is called at every phase change and every step."""

    def __init__(self, rank, limit):
        import threading
        self.rank, self.limit = rank, float(limit)
        self.phase, self.last = "start", time.monotonic()
        self.done = False
        self.fallback = None      # callable: what this rank still owes its caller when the watchdog fires (rank 0: the line already measured)
        if self.limit > 0:
            threading.Thread(target=self._run, name="dd-watchdog", daemon=True).start()

    def beat(self, phase):
        self.phase, self.last = phase, time.monotonic()

    def stop(self):
        self.done = True

    def _run(self):
        import faulthandler
        while not self.done:
            time.sleep(min(1.0, self.limit / 4))
            idle = time.monotonic() - self.last
            if not self.done and idle > self.limit:
                sys.stderr.write(f"bench.py watchdog: rank {self.rank} made no progress for {idle:.0f} s in phase '{self.phase}' "
                                 f"(limit {self.limit:.0f} s, DD_WATCHDOG_S); exiting with status 3\n")
                try:
                    if self.fallback is not None:
                        self.fallback()
                    faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                    sys.stderr.flush()
                finally:
                    os._exit(3)


class StdoutToStderr:
    """File-descriptor-level redirect of stdout into stderr for the span of communicator creation: with NCCL_DEBUG=VERSION (the GPU boxes
    of this pool export it) or WARN, RCCL prints a five-line version banner with printf on STDOUT the first time a communicator is
    created -- in front of the one JSON line this script owes its caller.  The user's NCCL_DEBUG stays as it is; the banner lands on stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def init_distributed(world, rank, dev, backend, dog):
    """Rendezvous with a BOUNDED wait, then one collective that every rank must reach (the preflight count)."""
    import datetime
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    limit = datetime.timedelta(seconds=float(os.environ.get("DD_DIST_TIMEOUT_S", "120")))
    dog.beat("rendezvous")
    with StdoutToStderr():      # RCCL's version banner (NCCL_DEBUG=VERSION / WARN) belongs on stderr
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=limit)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=limit)
        dog.beat("preflight collective")
        ones = torch.ones(1, device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        seen = int(round(float(ones.item())))
        if backend == "nccl":
            torch.cuda.synchronize()
    info = {"backend": backend, "n_ranks_seen": seen, "world_size": world, "dist_timeout_s": limit.total_seconds()}
    if dev is not None:
        info.update(visible_devices=torch.cuda.device_count(), device=torch.cuda.get_device_name(dev),
                    HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES"), ROCR_VISIBLE_DEVICES=os.environ.get("ROCR_VISIBLE_DEVICES"))
        if backend == "nccl":
            try:
                info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:      # noqa: BLE001  (diagnostics only)
                info["rccl_version"] = f"unavailable ({type(e).__name__})"
    if rank == 0:
        sys.stderr.write("bench.py preflight: " + json.dumps(info) + "\n")
        sys.stderr.flush()
    if seen != world:
        raise SystemExit(f"preflight: {seen} ranks answered, WORLD_SIZE is {world}")
    return info


def collective_probe(model, world, dev, backend, dog):
    """Measured bus bandwidth of the job's communicator: one all-reduce (SUM, fp32) over a buffer of the model's gradient size (648 MB
    for config 2; 16 MB over gloo, which is a rehearsal), 2 warm-up + 5 timed, max over ranks.  busbw = 2 (N - 1) / N x bytes / time,
    the figure RCCL's own tests quote; the step's collectives run with the same channel cap (NCCL_MAX_NCHANNELS) as this one."""
    import torch
    import torch.distributed as dist
    numel = sum(p.numel() for p in model.parameters())
    if backend != "nccl":
        numel = min(numel, 4 << 20)
    buf = torch.zeros(numel, device=dev if backend == "nccl" else "cpu", dtype=torch.float32)
    dog.beat("collective probe")
    for _ in range(2):
        dist.all_reduce(buf)
    if backend == "nccl":
        torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(5):
        dist.all_reduce(buf)
    if backend == "nccl":
        torch.cuda.synchronize()
    dt = torch.tensor([(time.perf_counter() - t0) / 5], dtype=torch.float64, device=buf.device)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    ms = float(dt.item()) * 1e3
    nbytes = numel * 4
    del buf
    return {"collective": "all_reduce fp32 sum", "message_bytes": nbytes, "ms": round(ms, 3),
            "algbw_GBps": round(nbytes / (ms * 1e-3) / 1e9, 1),
            "busbw_GBps": round(2.0 * (world - 1) / max(world, 1) * nbytes / (ms * 1e-3) / 1e9, 1)}


def fault_injection(rank, i, phase=""):
    """DD_BENCH_FAULT="<rank>:<step>[:hang[:default]]": that rank dies (or hangs) at that step -- the failure-path tests' lever.  With
    the fourth field only in the DEFAULT mode's steps (not in the all-reduce steps timed before them at N > 1)."""
    spec = os.environ.get("DD_BENCH_FAULT")
    if not spec:
        return
    parts = spec.split(":")
    if len(parts) > 3 and parts[3] == "default" and phase:
        return
    if int(parts[0]) == rank and int(parts[1]) == i:
        if len(parts) > 2 and parts[2] == "hang":
            sys.stderr.write(f"bench.py: injected hang on rank {rank} at step {i}\n")
            time.sleep(10 ** 6)
        sys.stderr.write(f"bench.py: injected fault on rank {rank} at step {i}\n")
        sys.stderr.flush()
        os._exit(17)


def control_flow_rehearsal(a, world, rank, dog):
    """DD_BENCH_CONTROL_ONLY=1: the rank's CONTROL flow -- bounded rendezvous, preflight, barriers, one collective per step,
    the timing reductions, the watchdog -- over gloo with no model and no kernel, for the failure-path tests on a machine without a
    GPU.  Prints a rehearsal record, never a benchmark line."""
    import torch
    import torch.distributed as dist
    info = init_distributed(world, rank, None, "gloo", dog)
    t = torch.zeros(1 << 16)
    dist.barrier()
    t0 = time.perf_counter()
    for i in range(a.warmup + a.steps):
        dog.beat(f"step {i}")
        fault_injection(rank, i)
        t.fill_(float(rank + i))
        dist.all_reduce(t)
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"control_flow_rehearsal": True, "n_ranks_seen": info["n_ranks_seen"], "steps": a.steps, "warmup": a.warmup}), flush=True)
    dog.stop()
    dist.destroy_process_group()


def run_rank(a):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        sys.stderr = RankStderr(rank, sys.stderr)
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE {world}: start either `python bench.py --gpus N` (self-launching) or "
                         f"torch.distributed.run with --nproc-per-node equal to --gpus")
    # the first import of torch on a fresh box pages the image in (1-2 min): the watchdog's clock starts after it
    import torch
    import torch.distributed as dist
    # rank 0's clock runs out first (x 0.8): when a job stalls, every rank's watchdog sees the same silence, and the launcher tears the
    # others down the moment the first one exits -- rank 0 has to have printed its fallback line (the all-reduce figure) by then
    dog = Watchdog(rank, float(os.environ.get("DD_WATCHDOG_S", "150")) * (0.8 if (rank == 0 and world > 1) else 1.0)
                   if world > 1 or os.environ.get("DD_WATCHDOG_S") else 0)
    if os.environ.get("DD_BENCH_CONTROL_ONLY") == "1":
        return control_flow_rehearsal(a, world, rank, dog)
    if world > 1:
        os.environ.setdefault("NCCL_MAX_NCHANNELS", os.environ.get("DD_RESERVED_CUS", "16"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # one process per GPU; DD_DIST_BACKEND=gloo + several ranks on ONE card is only a rehearsal of the N > 1
    # control flow on a single-GPU box (the real backend is nccl = RCCL over xGMI)
    backend = os.environ.get("DD_DIST_BACKEND", "nccl")
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    rehearse = world == 1 and os.environ.get("DD_REHEARSE_RCCL") == "1"      # the N > 1 call pattern on a 1-rank RCCL communicator
    if rehearse:
        os.environ.setdefault("NCCL_MAX_NCHANNELS", os.environ.get("DD_RESERVED_CUS", "16"))
        os.environ.setdefault("MASTER_PORT", "29533")
    comm = world > 1 or rehearse
    preflight = init_distributed(world, rank, dev, backend, dog) if comm else None

    from driving_dirty_amd import _lib
    from driving_dirty_amd import ops as _ops
    from driving_dirty_amd.train import TrainStep
    _lib.lib()                                            # fail loudly if the HIP library is missing
    # RCCL's collective workgroups run for milliseconds beside the conv backward and need LDS the conv workgroups do not
    # leave free: GradSync gives them their own compute units (and RCCL is capped at as many channels) from the first big
    # gradient to the end of the step, so the conv grids stay one resident round.  DD_RESERVED_CUS overrides.
    reserve = int(os.environ.get("DD_RESERVED_CUS", "16")) if comm else 0
    if a.cu_budget:
        _lib.check(_lib.lib().dd_set_cu_budget(a.cu_budget), "dd_set_cu_budget")

    globals().update(HIDDEN=a.hidden, LATENT=a.latent)
    if a.direct_conv:
        _ops.WINOGRAD = False
    if a.wino_1d:
        _ops.WINOGRAD_2D = False
    dog.beat("model")
    cfg = setup_config(a, dev, rank)
    model, batch, per_gpu = cfg["model"], cfg["batch"], cfg["per_gpu_batch"]
    overlap = {"on": True, "off": False, "auto": "auto"}[a.adam_overlap] if not a.no_adam_overlap else False      # auto: TrainStep decides by precision
    # sharded optimizer wherever the gradient message is big (tools/ddp_budget.py)
    shard = (comm or a.simulate_shard > 1) and {"on": True, "off": False, "auto": a.config in (2, 4, 5) or a.simulate_shard > 1}[a.shard_optimizer]
    # The optimizer and the gradient synchronisation are built over the model AS CONSTRUCTED -- feature extractor still frozen
    # (roadmap_bce_v2.py:45-47, spatial_w_rm.py:45-48): the first training_step unfreezes it where the config says so, and
    # LightningModule.unfreeze() re-arms both (driving_dirty_amd.train.TrainStep = HipAdam + GradSync, the step of this benchmark).
    # N = 2: the factors of the two big Linear layers instead of their gradients: replicated optimizer, no shards
    factor = comm and {"on": True, "off": False, "auto": world == 2 and a.config in (2, 4, 5) and a.shard_optimizer != "on"}[a.factor_linear]
    if factor:
        shard = False

    def make_step(shard_, factor_):
        return TrainStep(model, lr=1e-3, adam_overlap=overlap, shard_optimizer=shard_, reserve_cus=reserve, force_collectives=rehearse,
                         simulate_world=a.simulate_shard if a.simulate_shard > 1 else 0, scheduler=False, factor_linear=factor_,
                         fuse_linear_wgrad=a.fuse_linear_wgrad == "on", passes_last={"auto": "auto", "on": True, "off": False}[a.passes_last])

    def timed_region(ts, timer, tag):
        """W untimed steps, then EXACTLY K steps between barrier + synchronize on both sides; the MAX over ranks of the wall time."""
        def step(i):
            dog.beat(f"{tag}step {i}")
            fault_injection(rank, i, tag)
            return ts(batch, i)["loss"]
        for i in range(a.warmup):
            step(i)
        torch.cuda.synchronize()
        if comm:
            dog.beat(f"{tag}barrier before the timed region")
            dist.barrier()
        torch.cuda.synchronize()
        if timer is not None:
            timer.enabled = True
        t0 = time.perf_counter()
        for i in range(a.steps):
            loss = step(a.warmup + i)
        torch.cuda.synchronize()
        if comm:
            dog.beat(f"{tag}barrier after the timed region")
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if timer is not None:
            timer.enabled = False
        ts.sync_params()
        seen = 1
        if comm:
            dog.beat(f"{tag}timing reductions")
            cdev = dev if backend == "nccl" else "cpu"
            t = torch.tensor([dt], device=cdev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            ones = torch.ones(1, device=cdev, dtype=torch.float64)
            dist.all_reduce(ones, op=dist.ReduceOp.SUM)          # every rank that took part in the timed region counts itself
            seen = int(round(float(ones.item())))
        return dt, loss, seen

    # N > 1, first of all: the bus bandwidth of this communicator, and the PLAIN ALL-REDUCE step (replicated optimizer: the arrangement
    # every earlier round measured on one GPU) -- before the default mode, whose sharded optimizer / factor gather have met a multi-rank
    # RCCL communicator only in rehearsals.  Its figure goes to stderr at once and into the line (config.alt_all_reduce_ms); should the
    # default mode then stall, the watchdog prints a complete line for the all-reduce step before it exits, so the run is never empty.
    probe, alt = None, None
    if comm and not a.simulate_shard:
        probe = collective_probe(model, world, dev, backend, dog)
        if rank == 0:
            sys.stderr.write("bench.py collective probe: " + json.dumps(probe) + "\n")
            sys.stderr.flush()
    if comm and (shard or factor) and not a.simulate_shard and a.alt_all_reduce == "on":
        keep = {k: v.detach().clone() for k, v in model.state_dict().items()}
        rng = torch.cuda.get_rng_state(dev)
        ts0 = make_step(False, False)
        dt0, loss0, seen0 = timed_region(ts0, None, "all-reduce mode: ")
        ts0.close()
        del ts0
        alt = {"ms_per_step": round(dt0 / a.steps * 1e3, 3), "value": round(world * per_gpu * a.steps / dt0, 2), "n_ranks_seen": seen0,
               "final_loss": round(float(loss0.detach()), 6), "optimizer": "replicated (plain all-reduce of every gradient)"}
        if rank == 0:
            sys.stderr.write("bench.py alt_all_reduce: " + json.dumps(alt) + "\n")
            sys.stderr.flush()

            def fallback():      # the watchdog's last words on rank 0: a complete line for the mode that did run
                print(json.dumps({"metric": cfg["metric"], "value": alt["value"], "unit": "scenes/s", "n_gpus": world, "steps": a.steps,
                                  "warmup": a.warmup, "ms_per_step": alt["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                                  "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic", "n_ranks_seen": seen0,
                                  "config": {"workload": cfg["workload"], "baseline_config": a.config, "global_batch": world * per_gpu,
                                             "parallelism": f"dp{world}", "final_loss": alt["final_loss"], "optimizer": alt["optimizer"],
                                             "default_mode_failed": f"watchdog in phase '{dog.phase}'"},
                                  "roofline": None, "collective_probe": probe}), flush=True)
            dog.fallback = fallback
        # back to the state the default mode would have started from: parameters, BatchNorm statistics, the dropout generator (the
        # extractor stays unfrozen -- training_step switched it on in the all-reduce phase; the default mode hooks it from the start)
        with torch.no_grad():
            for k, v in model.state_dict().items():
                v.copy_(keep[k])
        del keep
        torch.cuda.set_rng_state(rng, dev)
        torch.cuda.empty_cache()

    ts = make_step(shard, factor)
    if a.adam_blocks_per_cu:
        _lib.check(_lib.lib().dd_set_adam_blocks_per_cu(a.adam_blocks_per_cu), "dd_set_adam_blocks_per_cu")
    if a.config == 2:
        timer = KernelTimer()
    else:
        timer = AbiTimer(cfg["watch"])
    timer.install()
    dt, loss, n_ranks_seen = timed_region(ts, timer, "")
    dog.fallback = None
    loss_val = float(loss.detach())
    dog.stop()                                            # what follows (rank 0's diagnostics, the CPU baseline) takes minutes by design

    if rank == 0:
        ms = dt / a.steps * 1e3
        value = world * per_gpu * a.steps / dt
        pmc = None
        if a.config == 2:
            roof, pmc = config2_roofline(timer, _ops)
        else:
            roof = watched_roofline(a.config, timer, per_gpu, ms)
        line = {
            "metric": cfg["metric"],
            "value": round(value, 2), "unit": "scenes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": cfg["dtype"], "data": "synthetic", "n_ranks_seen": n_ranks_seen,
            **({"rehearsal": "N > 1 call pattern on a 1-rank RCCL communicator (DD_REHEARSE_RCCL=1)"} if rehearse else {}),
            **({"simulated": f"COMPUTE side of a {a.simulate_shard}-GPU sharded step on one GPU: Adam on rank 0's 1/{a.simulate_shard} of every big "
                             "tensor, no collectives; a per-GPU lower bound, not a training step (parameters meaningless)"} if a.simulate_shard > 1 else {}),
            "config": {"workload": cfg["workload"], "baseline_config": a.config, "global_batch": world * per_gpu,
                       "parallelism": f"dp{world}", "final_loss": round(loss_val, 6),
                       "adam_overlap": bool(ts.overlap),
                       "linear_wgrad": ("formed inside the Adam pass (rank-B): " + ", ".join(sorted(n for n, p in model.named_parameters()
                                                                                                      if any(p is w for w in ts.fused))))
                       if ts.fused and not (shard or (comm and not factor)) else "materialised (dd_linear_wgrad + dd_adam_step)",
                       "optimizer": "sharded (reduce-scatter, Adam on 1/N, all-gather)" if shard else
                       ("replicated; the big Linear layers all-gather their factors, every rank forms the global-batch gradient" if factor
                        else "replicated")},
            # the step's algorithmic flops over its time, against the dense matrix peak of the dtype its convolutions run in
            ("step_algorithmic_frac_of_bf16_mfma_peak" if cfg["dtype"] == "bf16" else "step_algorithmic_frac_of_fp32_mfma_peak"):
                round(cfg["flop_per_scene"] * per_gpu * world / (ms * 1e-3) / 1e12
                      / ((PEAK_BF16_MFMA_TF if cfg["dtype"] == "bf16" else PEAK_F32_MFMA_TF) * world), 4),
            "roofline": roof,
        }
        if alt is not None:
            line["config"]["alt_all_reduce_ms"] = alt["ms_per_step"]
            line["config"]["alt_all_reduce"] = alt
        elif comm:
            line["config"]["alt_all_reduce_ms"] = round(ms, 3) if not (shard or factor) else None
        if probe is not None:
            line["collective_probe"] = probe
        if preflight is not None:
            line["preflight"] = preflight
        _phase("headline")
        if world == 1 and a.config == 2 and not a.no_others and not a.simulate_shard:
            ts.close()
            timer.uninstall()
            del model, ts, batch, cfg
            torch.cuda.empty_cache()
            line["others"] = other_configs(a, dev)
        if world == 1 and a.config == 2 and pmc is not None and not a.no_others and not a.simulate_shard:
            torch.cuda.empty_cache()      # this process's models are gone by now (deleted before `others`): the children get the GPU to themselves
            refresh_traffic(roof, pmc)
        _phase("traffic")
        if not a.no_cpu_baseline and world == 1 and a.config == 2 and not a.simulate_shard:
            line["cpu_baseline"] = cpu_baseline()
            _phase("cpu_baseline")
        print(json.dumps(line), flush=True)
    if comm:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))          # the parent: no torch import, no GPU call
    run_rank(a)


if __name__ == "__main__":
    main()
