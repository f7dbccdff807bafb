"""Build profiles/*_pmc_hbm.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py.

    python tests/tools/pmc_hbm.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json>

Per kernel: launches, mean FETCH_SIZE / WRITE_SIZE (KiB) per launch and HBM bytes per launch =
(2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM / rocprofv3 section).  Kernel names are reduced to the form bench.py's roofline uses."""
import collections, csv, glob, json, re, sys


def norm(name: str) -> str:
    """'void mopoe::gather_gemm_kernel<64, 64, 2, 2, 16, true, 3>(mopoe::GemmArgs)' -> 'gather_gemm_kernel<64, 64, 2, 2, 16, true, 3>'"""
    n = name.replace("void ", "").replace("mopoe::", "")
    depth = 0
    for i, ch in enumerate(n):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            n = n[:i]
            break
    return n[:90]


def collect(path, counter):
    """mean counter value per launch over the LAST complete train step of the run (the first step of a process also
    holds the autotuner's candidate launches); steps are delimited by latent_fwd_kernel, which runs once per step"""
    rows = []
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "latent_fwd_kernel" in r[1]]
    assert len(marks) >= 2, "need at least two train steps in the profiled run"
    agg = collections.defaultdict(lambda: [0, 0.0])
    for _, name, v in rows[marks[-2]:marks[-1]]:
        a = agg[norm(name)]
        a[0] += 1
        a[1] += v
    return agg


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `MOPOE_GRAPH=0 bench.py --steps 3 --warmup 1 "
               "--no-cpu-baseline --no-roofline`; hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE reports "
               "half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section); averages over all launches of the kernel "
               "(all layers it serves) in the last complete train step of the run", "kernels": {}}
for k, (n, tot) in sorted(fetch.items(), key=lambda kv: -kv[1][1]):
    w = write.get(k, [n, 0.0])
    fk, wk = tot / n, w[1] / max(w[0], 1)
    out["kernels"][k] = {"launches": n, "fetch_kb": round(fk, 1), "write_kb": round(wk, 1),
                         "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in list(out["kernels"].items())[:14]:
    print(k, v)
