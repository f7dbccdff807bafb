#!/bin/bash
# Where the bf16 implicit-GEMM kernels spend their cycles and what they fetch: register-staged tiles against the LDS-DMA
# tiles on C3's heaviest layer (rb1, 64 -> 128, k4 s2, B = 256).  Two rocprofv3 --pmc passes per (op, tile): SQ and TCC.
#   bash tests/tools/pmc_glds.sh > gpurun_out/pmc_glds.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for spec in "fwd 3" "fwd 7" "fwd 5" "dgrad 4" "dgrad 9" "dgrad 10" "dgrad 1"; do
  set -- $spec
  for pass in sq tcc; do
    rm -rf /tmp/pmc_g
    if [ $pass = sq ]; then
      C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
    else
      C="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
    fi
    rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_g -o r -- python3 $R/tests/tools/pmc_layer_bf16.py $1 $2 1 > /tmp/pmc_g.log 2>&1 || { echo "rocprofv3 failed for $spec $pass"; tail -5 /tmp/pmc_g.log; continue; }
    python3 - "$1" "$2" "$pass" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_g/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:78]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(f"{sys.argv[1]} tile {sys.argv[2]} [{sys.argv[3]}] {k}")
    wc = None
    if "SQ_WAVE_CYCLES" in d:
        wc = sum(d["SQ_WAVE_CYCLES"]) / len(d["SQ_WAVE_CYCLES"])
    for c, v in sorted(d.items()):
        m = sum(v) / len(v)
        print(f"   {c:28s} {m:16.0f}" + (f"  {m / wc:7.3f} of WAVE_CYCLES" if wc else ""))
    if "TCC_HIT_sum" in d:
        h, ms = sum(d["TCC_HIT_sum"]) / len(d["TCC_HIT_sum"]), sum(d["TCC_MISS_sum"]) / len(d["TCC_MISS_sum"])
        rd = sum(d["TCC_EA0_RDREQ_sum"]) / len(d["TCC_EA0_RDREQ_sum"])
        print(f"   L2 hit rate {h / max(h + ms, 1):.3f};  L2 requests x 128 B = {(h + ms) * 128 / 1e6:.0f} MB;  fabric reads x 64 B = {rd * 64 / 1e6:.0f} MB (x 2 for wide streaming reads: guide, HBM section)")
PY
  done
done
