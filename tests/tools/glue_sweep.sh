#!/bin/bash
# Sweep of the row-streaming glue kernels' launch knobs (tests/tools/glue_time.py) -> gpurun_out/glue_sweep.txt
# usage: bash tests/tools/glue_sweep.sh  (on a GPU box)
out=gpurun_out/glue_sweep.txt
mkdir -p gpurun_out
: > $out
for dt in fp32 bf16; do
  for u in ${UNROLLS:-1 4}; do
    for mb in ${MAXBLOCKS:-512 1024 2048}; do
      echo "== $dt unroll $u max_blocks $mb" >> $out
      MOPOE_EW_UNROLL=$u MOPOE_EW_MAX_BLOCKS=$mb python tests/tools/glue_time.py $dt 2>/dev/null | grep -E "^rows|^sum" >> $out || exit 1
    done
  done
done
grep -E "^==|^sum" $out
