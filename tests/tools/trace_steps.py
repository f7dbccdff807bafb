"""Summarise the steady-state steps of a rocprofv3 --kernel-trace CSV of bench.py (tuning aid):
per-kernel ms/step, sum of durations, union-busy time and idle gaps per step."""
import csv, glob, sys, collections
path = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r[1])
idx = [i for i, r in enumerate(rows) if "latent_fwd" in r[0]]
sub = rows[idx[-nsteps - 1]:idx[-1]]
span = (sub[-1][2] - sub[0][1]) / 1e6 / nsteps
byk = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in sub:
    k = n.split("(")[0].replace("void ", "").replace("mopoe::", "")[:58]
    byk[k][0] += 1; byk[k][1] += (e - s) / 1e3
tot = sum(v[1] for v in byk.values())
union, (cs, ce) = 0, sub[0][1:]
gaps = []
for _, s, e in sub[1:]:
    if s > ce:
        union += ce - cs; gaps.append(s - ce); cs, ce = s, e
    else:
        ce = max(ce, e)
union += ce - cs
print(f"kernels/step {len(sub)/nsteps:.0f}  span/step {span:.2f} ms  sum of durations/step {tot/1e3/nsteps:.2f} ms  union busy/step {union/1e6/nsteps:.2f} ms")
gaps.sort()
print(f"idle gaps/step {len(gaps)/nsteps:.0f}, total {sum(gaps)/1e6/nsteps:.2f} ms, median {gaps[len(gaps)//2]/1e3:.1f} us, p90 {gaps[int(len(gaps)*0.9)]/1e3:.1f} us")
for k, (c, t) in sorted(byk.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{k:60s} calls/step {c/nsteps:7.1f} ms/step {t/1e3/nsteps:7.3f} avg_us {t/c:7.1f}")
