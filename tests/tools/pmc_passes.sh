#!/bin/bash
# HBM bytes per kernel from the PMC counters (MI355X_MICROARCH.md, HBM / rocprofv3 section): two separate rocprofv3 --pmc
# passes (FETCH_SIZE, WRITE_SIZE; counters only, no trace domains) over the eager single-stream step of one config.
#   tests/tools/pmc_passes.sh c2   -> gpurun_out/pmc_hbm_<cfg>.json (+ the per-step total on stdout)
set -e
CFG=${1:-c2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${CFG}_$C
  MOPOE_GRAPH=0 MOPOE_NET_STREAMS=0 MOPOE_WGRAD_STREAM=0 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_${CFG}_$C -o r -- python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/pmc_${CFG}_$C.log 2>&1
done
python3 $R/tests/tools/pmc_hbm.py /tmp/pmc_${CFG}_FETCH_SIZE /tmp/pmc_${CFG}_WRITE_SIZE $R/gpurun_out/pmc_hbm_$CFG.json | head -16
python3 $R/tests/tools/pmc_total.py $R/gpurun_out/pmc_hbm_$CFG.json
