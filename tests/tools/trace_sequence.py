"""In-order kernel sequence of the last complete train step of a rocprofv3 --kernel-trace CSV (tuning aid: where do the
small launches of a step come from).   python tests/tools/trace_sequence.py <trace dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r[1])
idx = [i for i, r in enumerate(rows) if "latent_fwd" in r[0]]
sub = rows[idx[-2]:idx[-1]]
def short(n):
    n = n.replace("void ", "").replace("mopoe::", "").replace("at::native::", "").replace("(anonymous namespace)::", "")
    for a, b in (("vectorized_elementwise_kernel<4, ", "ew<"), ("std::array<char*, ", "arr"), ("elementwise_kernel_manual_unroll", "ewmu")):
        n = n.replace(a, b)
    return n[:90]
prev, cnt = None, 0
for n, s, e in sub:
    k = short(n)
    if k == prev:
        cnt += 1
        continue
    if prev is not None:
        print(f"{cnt:3d} x {prev}")
    prev, cnt = k, 1
print(f"{cnt:3d} x {prev}")
