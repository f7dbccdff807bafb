"""Which hardware queue each kernel of a replayed step ran on, and a time-ordered excerpt (tuning aid).
python tests/tools/queue_view.py <rocprof output dir> [first kernel index of the excerpt] [length]"""
import csv, glob, sys, collections
path = sys.argv[1]; a = int(sys.argv[2]) if len(sys.argv) > 2 else 0; n = int(sys.argv[3]) if len(sys.argv) > 3 else 80
f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    wg = max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    rows.append((r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mopoe::", "")[:52], int(r["Start_Timestamp"]),
                 int(r["End_Timestamp"]), grid // wg, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort(key=lambda r: r[1])
idx = [i for i, r in enumerate(rows) if "latent_fwd" in r[0]]
sub = rows[idx[-2]:idx[-1]]
q = collections.defaultdict(lambda: [0, 0.0])
for k, s, e, b, qi, si in sub:
    q[(qi, si)][0] += 1; q[(qi, si)][1] += (e - s) / 1e3
print("step of", len(sub), "kernels; per (queue, stream): kernels, summed duration us")
for key, (c, t) in sorted(q.items(), key=lambda kv: -kv[1][1]):
    print("  ", key, c, round(t, 1))
t0 = sub[0][1]
print("excerpt: start_us dur_us blocks queue stream kernel")
for k, s, e, b, qi, si in sub[a:a + n]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {b:6d} {qi:>3} {si:>3} {k}")
