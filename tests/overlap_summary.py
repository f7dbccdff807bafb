"""From a rocprofv3 --kernel-trace CSV: per-queue/stream kernel time, union busy time, overlap (tuning aid)."""
import collections, csv, glob, sys
path = sys.argv[1]
f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("columns:", list(rows[0].keys()))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows)
# keep the last 60 % of the trace (steady state)
t0 = iv[0][0] + (iv[-1][1] - iv[0][0]) * 0.4
iv = [x for x in iv if x[0] >= t0]
span = iv[-1][1] - iv[0][0]
tot = sum(e - s for s, e, _, _ in iv)
union, cur_s, cur_e = 0, None, None
for s, e, _, _ in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print(f"span {span/1e6:.2f} ms  sum of kernel durations {tot/1e6:.2f} ms  union busy {union/1e6:.2f} ms  idle {100*(1-union/span):.1f}%  overlap factor {tot/union:.3f}")
byq = collections.defaultdict(lambda: [0, 0])
for s, e, q, st in iv:
    byq[(q, st)][0] += 1; byq[(q, st)][1] += e - s
for k, (c, t) in sorted(byq.items(), key=lambda kv: -kv[1][1]):
    print(f"queue {k[0]} stream {k[1]}: {c} kernels {t/1e6:.2f} ms")
