#!/usr/bin/env python3
"""The 8-GPU budget of DESIGN.md section 6, computed from MEASURED one-GPU phases (profiles/r04_step_phases.json,
profiles/r04_simulated_shard8.json) and ASSUMED link speeds -- no multi-GPU hardware was available to the builder.

For each BASELINE config on 8 GPUs: message bytes; the window an all-reduce can hide under (big gradients ready -> end of backward);
the extra window the sharded optimizer's all-gather gets (step start -> first read of the big weight in the next forward); exposed
communication at three bus bandwidths (nccl-tests' definition: an all-reduce of S bytes takes 2(N-1)/N x S / busbw, a
reduce-scatter or all-gather (N-1)/N x S / busbw); projected step and efficiency T1 / T8.

    python tools/ddp_budget.py            # prints the table (markdown)
"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 8
BUSBW = (150.0, 330.0, 600.0)            # GB/s: one xGMI link's worth, what RCCL reaches on MI300-class nodes, near wire speed over 7 links
FIXED_MS = 0.32                          # 16 of 256 CUs handed to RCCL + the collective launches (profiles/r02_rccl_rehearsal.json)


def main():
    ph = json.load(open(os.path.join(ROOT, "profiles", "r04_step_phases.json")))
    sim = json.load(open(os.path.join(ROOT, "profiles", "r04_simulated_shard8.json")))
    t1 = {2: min(sim["sim_c2_s0"]["ms_per_step"], sim["sim2_c2_s0"]["ms_per_step"]), 5: min(sim["sim_c5_s0"]["ms_per_step"], sim["sim2_c5_s0"]["ms_per_step"])}
    t8c = {2: min(sim["sim_c2_s8"]["ms_per_step"], sim["sim2_c2_s8"]["ms_per_step"]), 5: min(sim["sim_c5_s8"]["ms_per_step"], sim["sim2_c5_s8"]["ms_per_step"])}
    rows = []
    for c in (2, 4, 5):
        p = ph[f"config{c}"]
        m = p["ms_from_step_start"]
        s_gb = sum(p["big_tensors_MB"].values()) / 1e3
        ready = max(v for k, v in m.items() if k.startswith("grad_ready"))
        w_bwd = m["backward_end"] - ready
        w_fwd = min(v for k, v in m.items() if k.startswith("first_read"))
        step1 = t1.get(c, m["step_end"])
        comp8 = t8c.get(c, step1 - 0.4)          # config 4: the sharded Adam's gain measured on config 2 (0.4 ms), same tensors
        for mode in ("all-reduce", "sharded"):
            cells = []
            for bw in BUSBW:
                t_ar = 2.0 * (N - 1) / N * s_gb / bw * 1e3
                if mode == "all-reduce":
                    exposed = max(0.0, t_ar - w_bwd) + 0.35          # + the Adam pass of the last 128 MB piece
                    t8 = step1 + exposed + FIXED_MS
                else:
                    # reduce-scatter under the backward; the all-gathers start behind each piece's Adam, still inside the backward, and
                    # must land before the next forward first reads the weight
                    t_half = t_ar / 2.0
                    exposed = max(0.0, 2.0 * t_half - (w_bwd + w_fwd))
                    t8 = comp8 + exposed + FIXED_MS
                cells.append((exposed, t8, step1 / t8))
            rows.append((c, mode, s_gb, w_bwd, w_fwd, step1, comp8, cells))
    print("| config | optimizer | message | hide window: backward / + next forward | 1-GPU step | compute side at 8 | exposed ms @150 / 330 / 600 GB/s | 8-GPU step ms | efficiency |")
    print("|---|---|---|---|---|---|---|---|---|")
    for c, mode, s_gb, w_bwd, w_fwd, step1, comp8, cells in rows:
        print(f"| {c} | {mode} | {s_gb:.2f} GB | {w_bwd:.1f} / +{w_fwd:.1f} ms | {step1:.2f} | {(comp8 if mode == 'sharded' else step1):.2f} | "
              + " / ".join(f"{e:.1f}" for e, _, _ in cells) + " | " + " / ".join(f"{t:.1f}" for _, t, _ in cells) + " | "
              + " / ".join(f"{f:.2f}" for _, _, f in cells) + " |")


def curve():
    """The driver's N = 1, 2, 4, 8 curve for config 2 under a per-link model: xGMI is point-to-point, a GPU talks to each of its
    N - 1 peers over one link of ~76.8 GB/s per direction (153.6 GB/s bidirectional), so the bus bandwidth a direct reduce-scatter /
    all-gather can reach GROWS with N: (N - 1) x 76.8 GB/s at wire speed; `eff` of it is assumed delivered.  N = 2 is the hard
    point: one link carries the whole message."""
    ph = json.load(open(os.path.join(ROOT, "profiles", "r04_step_phases.json")))["config2"]
    sim = json.load(open(os.path.join(ROOT, "profiles", "r04_simulated_shard8.json")))
    m = ph["ms_from_step_start"]
    s_gb = sum(ph["big_tensors_MB"].values()) / 1e3
    w_bwd = m["backward_end"] - max(v for k, v in m.items() if k.startswith("grad_ready"))
    w_fwd = min(v for k, v in m.items() if k.startswith("first_read"))
    t1 = min(sim["sim_c2_s0"]["ms_per_step"], sim["sim2_c2_s0"]["ms_per_step"])
    t_shard8 = min(sim["sim_c2_s8"]["ms_per_step"], sim["sim2_c2_s8"]["ms_per_step"])
    # factor gather (ddp.GradSync(factor_linear=True)): every rank receives (N - 1) x 202 MB -- fc1's 120 MB of pooled activations, on the
    # links from the end of the forward's conv part, and the head's 82 MB of dlogits from the top of the backward -- and needs them in
    # front of c2's weight gradient, which goes LAST in this mode (ops.C2_DGRAD_FIRST).  Compute side measured on a one-rank RCCL
    # communicator: +0.2 ms over the plain step (the two weight-gradient launches over N x 32 rows, the reordered backward).
    f_gb = 0.202
    w_fac = m["backward_end"] - 1.4 - m["forward_end"] + 0.5      # forward tail (0.5 ms) + the backward up to c2's weight gradient (its last ~1.4 ms)
    print()
    print("| N | busbw at 70 % of (N-1) links | all-reduce: exposed / step / speed-up | sharded: exposed / step / speed-up | factor gather: exposed / step / speed-up |")
    print("|---|---|---|---|---|")
    for n in (2, 4, 8):
        bw = 0.7 * (n - 1) * 76.8
        t_ar = 2.0 * (n - 1) / n * s_gb / bw * 1e3
        e_ar = max(0.0, t_ar - w_bwd) + 0.35
        step_ar = t1 + e_ar + FIXED_MS
        comp = t1 - (t1 - t_shard8) * (1.0 - 1.0 / n) / (1.0 - 1.0 / 8)      # the Adam saving scales with the share given away
        e_sh = max(0.0, t_ar - (w_bwd + w_fwd))
        step_sh = comp + e_sh + FIXED_MS
        t_fac = (n - 1) * f_gb / bw * 1e3
        e_fac = max(0.0, t_fac - w_fac)
        step_fac = t1 + 0.2 + 0.08 * (n - 2) + e_fac + FIXED_MS                # the weight-gradient launches grow with the gathered batch
        print(f"| {n} | {bw:.0f} GB/s | {e_ar:.1f} / {step_ar:.1f} ms / {n * t1 / step_ar:.2f}x | {e_sh:.1f} / {step_sh:.1f} ms / {n * t1 / step_sh:.2f}x | "
              f"{e_fac:.1f} / {step_fac:.1f} ms / {n * t1 / step_fac:.2f}x |")


if __name__ == "__main__":
    main()
    curve()
