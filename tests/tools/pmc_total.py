"""Per-step HBM traffic by kernel family from a pmc_hbm json (tests/tools/pmc_hbm.py): sum of launches x bytes per launch."""
import collections, json, sys
d = json.load(open(sys.argv[1]))["kernels"]
fam = collections.defaultdict(float)
tot = 0.0
for k, v in d.items():
    b = v["launches"] * v["hbm_bytes_per_launch"]
    tot += b
    name = k.split("<")[0]
    fam[name] += b
print(f"total {tot / 1e9:.2f} GB/step")
for k, b in sorted(fam.items(), key=lambda kv: -kv[1])[:14]:
    print(f"  {k:40s} {b / 1e9:7.2f} GB")
