#!/usr/bin/env python3
"""Measure the launch plans of the BASELINE configurations on an MI355X and write the table the product commits
(mopoe-mimic_amd/mimic_amd/plans_gfx950.json; VERDICT r3 item 5: deterministic launch plans).

    python tests/tools/make_plan_table.py --out gpurun_out/plans_gfx950.json [--configs c2 c3 c5 c2d128 c1] [--merge FILE]

Each configuration runs in a fresh child process (one GPU user at a time): the model of bench.py, train mode with dropout,
the tuner forced on (MOPOE_AUTOTUNE=force: the committed table is ignored) with longer timing batches than the in-process
default, two eager train steps (every (op, geometry, fusion) triple of a step is met in the first), then the tuner's table is
dumped.  The parent merges the children's tables; a triple two configurations share keeps the first one's plan."""
import argparse
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "mopoe-mimic_amd")]


def child(cfg_name, out):
    import torch
    import bench
    from mimic_amd import ops, run_epochs as RE
    from mimic_amd.utils.experiment import HotPathExperiment, default_flags
    size, cdim, dimg, bsz, cdtype = bench.CONFIGS[cfg_name]
    device = torch.device("cuda", 0)
    torch.manual_seed(0)
    flags = default_flags(img_size=size, class_dim=cdim, DIM_img=dimg, batch_size=bsz, device=device,
                          initial_learning_rate=1e-5, compute_dtype=cdtype)
    exp = HotPathExperiment(flags)
    exp.mm_vae.to(device)
    exp.mm_vae.train()
    exp.set_optimizer()
    batches = bench.synthetic_batches(flags, 2, device, seed=1)
    pack = RE.ScalarPack(device)
    # sustained load first: a GPU that has just left idle ranks candidates differently from steady state
    xs = torch.randn(4096, 4096, device=device)
    for _ in range(200):
        xs = torch.tanh(xs @ xs * 1e-3)
    torch.cuda.synchronize()
    for b in batches:
        RE.train_step(exp, ({k: v for k, v in b[0].items()}, None), None, pack)
    torch.cuda.synchronize()
    plans = {ops.plan_key_str(k): (None if v is None else list(v)) for k, v in ops.plan_table().items()}
    report = {ops.plan_key_str((r[0], r[1]) + tuple(r[2])): {"chosen_us": r[4], "candidates": {f"{c[0]},{c[1]}": round(t, 2) for c, t in r[5].items()}}
              for r in ops.plan_report()}
    with open(out, "w") as f:
        json.dump({"plans": plans, "report": report}, f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", nargs="*", default=["c2", "c3", "c5", "c2d128", "c1"])
    ap.add_argument("--out", default="gpurun_out/plans_gfx950.json")
    ap.add_argument("--merge", default=None, help="an existing table whose entries are kept for triples not measured now")
    ap.add_argument("--child", default=None)
    args = ap.parse_args()
    if args.child:
        child(args.child, args.out)
        return
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    plans, reports = {}, {}
    for cfg in args.configs:
        tmp = f"{args.out}.{cfg}.json"
        env = dict(os.environ, MOPOE_AUTOTUNE="force", MOPOE_TUNE_REPS=os.environ.get("MOPOE_TUNE_REPS", "6"))
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", cfg, "--out", tmp], check=True, env=env)
        with open(tmp) as f:
            d = json.load(f)
        new = 0
        for k, v in d["plans"].items():
            if k not in plans:
                plans[k] = v
                new += 1
        reports[cfg] = d["report"]
        print(f"[plans] {cfg}: {len(d['plans'])} triples, {new} new", flush=True)
        os.remove(tmp)
    configs = list(args.configs)
    if args.merge and os.path.exists(args.merge):
        with open(args.merge) as f:
            old = json.load(f)
        for k, v in old.get("plans", {}).items():
            plans.setdefault(k, v)
        configs += [c for c in old.get("meta", {}).get("configs", []) if c not in configs]   # the kept triples' configurations
    import torch
    meta = {"device": "MI355X (gfx950)", "torch": torch.__version__, "configs": configs,
            "note": "written by tests/tools/make_plan_table.py; key = op|geometry(N,Hs,Ws,Hb,Wb,Cin,Cout,kh,kw,sh,sw,ph,pw,transposed)|fusion flags; "
                    "value = [tile, split] (include/mopoe_hip.h: mopoe_conv_plan) or null = the library's static heuristic"}
    with open(args.out, "w") as f:
        json.dump({"meta": meta, "plans": dict(sorted(plans.items()))}, f, indent=0)
    with open(args.out.replace(".json", "_report.json"), "w") as f:
        json.dump(reports, f)
    print(f"[plans] wrote {args.out}: {len(plans)} triples", flush=True)


if __name__ == "__main__":
    main()
