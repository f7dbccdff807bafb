#!/bin/bash
# Regenerates the measurements kept under profiles/ (run on the GPU box from the repo root; outputs go to gpurun_out/,
# copy the ones to keep into profiles/):
#   1. bench.py with its defaults                      -> gpurun_out/bench.json
#   2. rocprofv3 --kernel-trace --stats of bench.py    -> gpurun_out/kernel_stats.csv, kernel_summary.txt
#   3. the same with the eager single-stream step      -> gpurun_out/kernel_stats_eager_serial.csv, kernel_summary_eager_serial.txt
#   4. per-layer replay of every conv op               -> gpurun_out/layers_c2.txt
#   5. (separate call) PMC passes: see tests/tools/pmc_hbm.py
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python3 $R/bench.py > $R/gpurun_out/bench.json 2> $R/gpurun_out/bench.err
tail -c 2500 $R/gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/prof_stats.log 2>&1
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/kernel_stats.csv
python3 $R/tests/tools/trace_steps.py /tmp/prof_stats 6 60 > $R/gpurun_out/kernel_summary.txt
head -12 $R/gpurun_out/kernel_summary.txt
# the regime bench.py's roofline pass times kernels in: eager, one stream (kernel durations without co-runners)
rm -rf /tmp/prof_serial
MOPOE_GRAPH=0 MOPOE_NET_STREAMS=0 MOPOE_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_serial -o r -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/prof_serial.log 2>&1
cp $(find /tmp/prof_serial -name "*kernel_stats.csv" | head -1) $R/gpurun_out/kernel_stats_eager_serial.csv
python3 $R/tests/tools/trace_steps.py /tmp/prof_serial 6 60 > $R/gpurun_out/kernel_summary_eager_serial.txt
head -8 $R/gpurun_out/kernel_summary_eager_serial.txt
cd $R && python3 tests/tools/layer_replay.py c2 > gpurun_out/layers_c2.txt 2>&1
head -3 gpurun_out/layers_c2.txt
