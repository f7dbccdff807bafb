# Round-4 final measurement pass (fp32 products on the bf16 matrix pipe, tuner objective alpha 0.7).  One GPU box, ~6 min.
#   bash tests/tools/r04_measure2.sh [bench|prof|all]     -> gpurun_out/r04_*
set -x
W=${1:-all}
if [ $W = bench ] || [ $W = all ]; then
python bench.py > gpurun_out/r04_bench_c2.json 2> gpurun_out/r04_bench_c2.err
for c in c3 c5 c2d128; do python bench.py --config $c --no-cpu-baseline > gpurun_out/r04_bench_$c.json 2> gpurun_out/r04_bench_$c.err; done
MOPOE_FORCE_DP=1 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r04_bench_c2_forced_dp.json 2> gpurun_out/r04_bench_c2_forced_dp.err
MOPOE_F32_SPLIT_BF16=0 python bench.py --no-cpu-baseline > gpurun_out/r04_bench_c2_fp32_mfma_only.json 2> gpurun_out/r04_bench_c2_fp32_mfma_only.err
python tests/tools/wgrad9_time.py 64 > gpurun_out/r04_wgrad_parity_f32_time.txt 2>&1
python tests/tools/emu_probe.py 64 > gpurun_out/r04_emu_probe.txt 2>&1
fi
if [ $W = prof ] || [ $W = all ]; then
for c in c2 c3 c5; do
  tests/tools/profile_config.sh $c > gpurun_out/r04_prof_$c.log 2>&1; for f in kernel_summary_$c.txt kernel_summary_${c}_eager_serial.txt kernel_stats_$c.csv layers_$c.txt; do cp gpurun_out/$f gpurun_out/r04_$f; done
  python tests/tools/net_timeline.py $c > gpurun_out/r04_net_timeline_$c.txt 2>/dev/null
  tests/tools/pmc_passes.sh $c > gpurun_out/r04_pmc_$c.log 2>&1; cp gpurun_out/pmc_hbm_$c.json gpurun_out/r04_pmc_hbm_$c.json
done
fi
grep -h value gpurun_out/r04_bench_*.json | cut -c1-120
