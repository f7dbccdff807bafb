set -x
python tests/tools/make_plan_table.py --out gpurun_out/plans_gfx950.json --configs c2 c3 c5 c2d128 c1 > gpurun_out/r4_plans.log 2>&1
cp gpurun_out/plans_gfx950.json mopoe-mimic_amd/mimic_amd/plans_gfx950.json
python bench.py > gpurun_out/r04_bench_c2.json 2> gpurun_out/r04_bench_c2.err
for c in c3 c5 c2d128; do python bench.py --config $c --no-cpu-baseline > gpurun_out/r04_bench_$c.json 2> gpurun_out/r04_bench_$c.err; done
MOPOE_FORCE_DP=1 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r04_bench_c2_forced_dp.json 2> gpurun_out/r04_bench_c2_forced_dp.err
for c in c2 c3 c5; do tests/tools/profile_config.sh $c > gpurun_out/r04_prof_$c.log 2>&1; for f in kernel_summary_$c.txt kernel_summary_${c}_eager_serial.txt kernel_stats_$c.csv layers_$c.txt; do cp gpurun_out/$f gpurun_out/r04_$f; done; done
for c in c2 c3 c5; do python tests/tools/net_timeline.py $c > gpurun_out/r04_net_timeline_$c.txt 2>/dev/null; done
KINDS=wgrad python tests/tools/plan_time_bf16.py > gpurun_out/r04_wgrad_parity_time.txt 2>&1
python tests/tools/front_time.py > gpurun_out/r04_front_time.txt 2>&1
DTYPE=f32 python tests/tools/front_time.py >> gpurun_out/r04_front_time.txt 2>&1
grep -h value gpurun_out/r04_bench_*.json | cut -c1-160
