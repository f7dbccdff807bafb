#!/usr/bin/env python3
"""Refine a launch-plan table INSIDE the step: for the triples that cost the most, try the tuner's runner-up plans one at a
time and keep a change only if the replayed train step (one hipGraph, three networks' kernels side by side) gets faster.

    python tests/tools/refine_plan_table.py --config c2 --table mopoe-mimic_amd/mimic_amd/plans_gfx950.json \\
        --report gpurun_out/plans_report.json --out gpurun_out/plans_refined.json [--top 40] [--alts 3]

The isolated timings that built the table (tests/tools/make_plan_table.py) rank a plan by its own duration (x the share of the
chip it occupies); what a plan costs the STEP also depends on what runs beside it.  Greedy, one triple at a time, most expensive
first; a candidate is accepted when the step is faster by more than --gain (default 0.3 %) in two consecutive measurements."""
import argparse
import gc
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "mopoe-mimic_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--table", required=True)
    ap.add_argument("--report", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--skip", type=int, default=0, help="leave out the N costliest triples (examined by an earlier pass)")
    ap.add_argument("--alts", type=int, default=3)
    ap.add_argument("--gain", type=float, default=0.003)
    ap.add_argument("--steps", type=int, default=80)
    args = ap.parse_args()
    os.environ["MOPOE_PLAN_TABLE"] = os.path.abspath(args.table)
    os.environ["MOPOE_AUTOTUNE"] = "table"
    import torch
    import bench
    from mimic_amd import ops, run_epochs as RE
    from mimic_amd.utils.experiment import HotPathExperiment, default_flags
    size, cdim, dimg, bsz, cdtype = bench.CONFIGS[args.config]
    device = torch.device("cuda", 0)
    torch.manual_seed(0)
    flags = default_flags(img_size=size, class_dim=cdim, DIM_img=dimg, batch_size=bsz, device=device,
                          initial_learning_rate=1e-5, compute_dtype=cdtype)
    exp = HotPathExperiment(flags)
    exp.mm_vae.to(device)
    exp.mm_vae.train()
    exp.set_optimizer()
    batches = bench.synthetic_batches(flags, 4, device, seed=1)
    pack = RE.ScalarPack(device)
    with open(args.table) as f:
        table = json.load(f)
    with open(args.report) as f:
        report = json.load(f)[args.config]
    ops._table_plan(("wgrad", ops.Geom(1, 1, 1, 1, 1, 4, 4, 1, 1, 1, 1, 0, 0, False), False))     # loads the table into ops._plan_table

    def step_ms():
        ops.clear_plans()
        step = RE.GraphedTrainStep(exp, batches[0], pack, None)
        for i in range(10):
            step(batches[i % 4])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.steps):
            step(batches[i % 4])
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.steps
        del step
        gc.collect()
        torch.cuda.empty_cache()
        return ms

    base = min(step_ms(), step_ms())
    print(f"[refine] {args.config}: baseline {base:.4f} ms/step", flush=True)
    keys = [k for k in report if report[k].get("candidates") and k in ops._plan_table and ops._plan_table[k] is not None]
    keys.sort(key=lambda k: -(report[k].get("chosen_us") or 0.0))
    changed = {}
    for k in keys[args.skip:args.skip + args.top]:
        cur = tuple(ops._plan_table[k])
        cands = sorted(report[k]["candidates"].items(), key=lambda kv: kv[1])
        best_us = cands[0][1]
        alts = [tuple(int(x) for x in c.split(",")) for c, t in cands if t <= 1.35 * best_us]
        alts = [a for a in alts if a != cur][:args.alts]
        for alt in alts:
            ops._plan_table[k] = list(alt)
            t = step_ms()
            if t < base * (1.0 - args.gain):
                t2 = step_ms()
                if t2 < base * (1.0 - args.gain):
                    print(f"[refine] {k}: {cur} -> {alt}: {base:.4f} -> {max(t, t2):.4f} ms", flush=True)
                    base, cur = max(t, t2), alt
                    changed[k] = list(alt)
                    continue
            ops._plan_table[k] = list(cur)
    print(f"[refine] {args.config}: {len(changed)} triples changed, {base:.4f} ms/step", flush=True)
    table["plans"].update(changed)
    table["meta"]["refined_in_step"] = sorted(set(table["meta"].get("refined_in_step", []) + [args.config]))
    with open(args.out, "w") as f:
        json.dump({"meta": table["meta"], "plans": dict(sorted(table["plans"].items()))}, f, indent=0)
    print(f"[refine] wrote {args.out}", flush=True)


if __name__ == "__main__":
    main()
