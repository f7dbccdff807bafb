for envs in "X=1" "MOPOE_NET_STREAM_SET=PA,Lateral,text" "MOPOE_BLOCK_FRONT_F32=0" "MOPOE_BLOCK_FRONT_F32_FWD=1" "MOPOE_LAZY_HEAD=0" "X=2" "MOPOE_WGRAD_STREAM=0" "MOPOE_FUSE_MIX=0"; do
  env $envs python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$envs', d['value'], d['ms_per_step'])"
done
