#!/bin/bash
# rocprofv3 kernel trace of bench.py for one config (graphed step, and the eager single-stream regime) + per-layer replay.
#   tests/tools/profile_config.sh c3   -> gpurun_out/kernel_summary_<cfg>.txt, kernel_summary_<cfg>_eager_serial.txt, layers_<cfg>.txt
set -e
CFG=${1:-c2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$CFG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$CFG -o r -- python3 $R/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/prof_$CFG.log 2>&1
cp $(find /tmp/prof_$CFG -name "*kernel_stats.csv" | head -1) $R/gpurun_out/kernel_stats_$CFG.csv
python3 $R/tests/tools/trace_steps.py /tmp/prof_$CFG 6 60 > $R/gpurun_out/kernel_summary_$CFG.txt
head -40 $R/gpurun_out/kernel_summary_$CFG.txt
rm -rf /tmp/prof_${CFG}_s
MOPOE_GRAPH=0 MOPOE_NET_STREAMS=0 MOPOE_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${CFG}_s -o r -- python3 $R/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/prof_${CFG}_s.log 2>&1
python3 $R/tests/tools/trace_steps.py /tmp/prof_${CFG}_s 6 60 > $R/gpurun_out/kernel_summary_${CFG}_eager_serial.txt
head -45 $R/gpurun_out/kernel_summary_${CFG}_eager_serial.txt
cd $R && python3 tests/tools/layer_replay.py $CFG > gpurun_out/layers_$CFG.txt 2>&1
head -60 gpurun_out/layers_$CFG.txt
