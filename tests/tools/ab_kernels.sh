#!/bin/bash
# A/B of two environment settings in the eager single-stream regime: per-kernel ms/step of both, side by side.
#   tests/tools/ab_kernels.sh c3 "MOPOE_LAZY_HEAD=0" "MOPOE_LAZY_HEAD=1"   -> gpurun_out/ab_<cfg>.txt
CFG=${1:-c3}; A="$2"; B="$3"
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for side in A B; do
  if [ $side = A ]; then ENVS="$A"; else ENVS="$B"; fi
  rm -rf /tmp/ab_$side
  env $ENVS MOPOE_GRAPH=0 MOPOE_NET_STREAMS=0 MOPOE_WGRAD_STREAM=0 true
  ( export $ENVS MOPOE_GRAPH=0 MOPOE_NET_STREAMS=0 MOPOE_WGRAD_STREAM=0; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$side -o r -- python3 $R/bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/ab_$side.log 2>&1 )
  python3 $R/tests/tools/trace_steps.py /tmp/ab_$side 6 80 > $R/gpurun_out/ab_${CFG}_$side.txt
done
python3 - <<PY
import re
def load(p):
    d={}
    for line in open(p):
        m=re.match(r"(.{60})\s+calls/step\s+([\d.]+)\s+ms/step\s+([\d.]+)\s+avg_us\s+([\d.]+)", line)
        if m: d[m.group(1).strip()]=(float(m.group(2)),float(m.group(3)),float(m.group(4)))
    return d
a=load("$R/gpurun_out/ab_${CFG}_A.txt"); b=load("$R/gpurun_out/ab_${CFG}_B.txt")
rows=[]
for k in set(a)|set(b):
    xa=a.get(k,(0,0,0)); xb=b.get(k,(0,0,0))
    rows.append((xb[1]-xa[1],k,xa,xb))
rows.sort()
with open("$R/gpurun_out/ab_${CFG}.txt","w") as f:
    f.write("A: $A\nB: $B\n")
    f.write(f"total ms/step: A {sum(v[1] for v in a.values()):.3f}  B {sum(v[1] for v in b.values()):.3f}\n")
    for d,k,xa,xb in rows:
        if abs(d)>0.004:
            f.write(f"{d:+8.3f} ms  {k:60s} A {xa[0]:5.1f} x {xa[2]:7.1f}us = {xa[1]:.3f}   B {xb[0]:5.1f} x {xb[2]:7.1f}us = {xb[1]:.3f}\n")
print(open("$R/gpurun_out/ab_${CFG}.txt").read())
PY
