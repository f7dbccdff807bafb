"""Summarise a rocprofv3 --kernel-trace CSV: time per kernel and per (kernel, grid) in ms/step."""
import collections, csv, glob, sys
path = sys.argv[1]; steps = float(sys.argv[2])
f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
byk = collections.defaultdict(lambda: [0, 0.0]); byg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mopoe::", "")[:60]
    byk[n][0] += 1; byk[n][1] += d
    if "gemm" in n or "mopoe" in r["Kernel_Name"]:
        key = (n, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        byg[key][0] += 1; byg[key][1] += d
tot = sum(v[1] for v in byk.values())
print(f"total GPU ms/step {tot/1e3/steps:.3f}")
for k, (c, t) in sorted(byk.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{k:62s} calls/step {c/steps:7.1f}  ms/step {t/1e3/steps:8.3f}  avg_us {t/c:8.1f}")
print("---- by grid")
for k, (c, t) in sorted(byg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{str(k):80s} calls/step {c/steps:5.1f}  ms/step {t/1e3/steps:7.3f}  avg_us {t/c:8.1f}")
