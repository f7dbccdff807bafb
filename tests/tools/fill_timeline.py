"""How full the chip is over one replayed step (tuning aid): from a rocprofv3 --kernel-trace CSV of bench.py, the time
spent with 0 / 1 / 2 / 3+ kernels resident, the time during which all resident kernels together have fewer than 256
workgroups ("under-filled": a 256-CU chip cannot be busy), and which kernels that time belongs to.
python tests/tools/fill_timeline.py <rocprof output dir> [steps]"""
import csv, glob, sys, collections
path = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    wg = max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), grid // wg))
rows.sort(key=lambda r: r[1])
idx = [i for i, r in enumerate(rows) if "latent_fwd" in r[0]]
sub = rows[idx[-nsteps - 1]:idx[-1]]
ev = []
for i, (n, s, e, b) in enumerate(sub):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
live = set()
by_count = collections.Counter()
under = 0
under_by = collections.Counter()
under_by_n = collections.Counter()
prev = ev[0][0]
for t, d, i in ev:
    dt = t - prev
    if dt > 0:
        by_count[min(len(live), 3)] += dt
        blocks = sum(sub[j][3] for j in live)
        if live and blocks < 256:
            under += dt
            for j in live:
                k = sub[j][0].split("(")[0].replace("void ", "").replace("mopoe::", "")[:60]
                under_by[k] += dt / len(live)
                under_by_n[k] += 1
    prev = t
    if d > 0: live.add(i)
    else: live.discard(i)
span = (sub[-1][2] - sub[0][1])
print(f"span/step {span / 1e6 / nsteps:.2f} ms; resident kernels 0: {by_count[0] / 1e6 / nsteps:.2f}  1: {by_count[1] / 1e6 / nsteps:.2f}  "
      f"2: {by_count[2] / 1e6 / nsteps:.2f}  3+: {by_count[3] / 1e6 / nsteps:.2f} ms/step")
print(f"under-filled (< 256 resident workgroups in total): {under / 1e6 / nsteps:.2f} ms/step, by kernel:")
for k, t in under_by.most_common(25):
    print(f"  {k:62s} {t / 1e6 / nsteps:7.3f} ms/step")
# phases: under-filled time along the step in 20 slices
t0 = sub[0][1]
sl = [0.0] * 20
live, prev = set(), ev[0][0]
for t, d, i in ev:
    dt = t - prev
    if dt > 0 and live and sum(sub[j][3] for j in live) < 256:
        sl[min(19, int((prev - t0) * 20 / span))] += dt
    prev = t
    if d > 0: live.add(i)
    else: live.discard(i)
print("under-filled share per 5 % slice of the traced span:", " ".join(f"{x / (span / 20):.2f}" for x in sl))
