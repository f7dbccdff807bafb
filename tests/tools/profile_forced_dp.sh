set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_dp
MOPOE_FORCE_DP=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dp -o r -- python3 $R/bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/prof_dp.log 2>&1
python3 $R/tests/tools/trace_steps.py /tmp/prof_dp 6 60 > $R/gpurun_out/kernel_summary_c2_forced_dp.txt
head -12 $R/gpurun_out/kernel_summary_c2_forced_dp.txt
grep -i "rccl\|nccl\|AllReduce\|Broadcast" $R/gpurun_out/kernel_summary_c2_forced_dp.txt | head
