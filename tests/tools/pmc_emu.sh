#!/bin/bash
# SQ counters of the fp32 LDS-DMA tiles with the products on the fp32 MFMA (15, 13) and on the bf16 matrix pipe (16, 19, 17) on
# C2's heaviest layer (rb1, 64 -> 128, k4 s2, B = 64): who keeps the SIMDs busy.
#   bash tests/tools/pmc_emu.sh ["op tile split" ...] > gpurun_out/pmc_emu.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
if [ $# -eq 0 ]; then set -- "fwd 15 1" "fwd 16 1" "fwd 19 1" "dgrad 13 1" "dgrad 17 1" "wgrad 2 64" "wgrad 9 32" "wgrad 10 64"; fi
for spec in "$@"; do
  set -- $spec
  rm -rf /tmp/pmc_g
  C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
  rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_g -o r -- python3 $R/tests/tools/pmc_layer_f32.py $1 $2 $3 > /tmp/pmc_g.log 2>&1 || { echo "rocprofv3 failed for $spec"; tail -5 /tmp/pmc_g.log; continue; }
  python3 - "$1" "$2" "$3" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmc_g/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"] or "parity" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(f"{sys.argv[1]} tile {sys.argv[2]} split {sys.argv[3]}: {k}")
    for c, v in sorted(d.items()):
        print(f"   {c:30s} {sum(v) / len(v):16.0f}")
PY
done
