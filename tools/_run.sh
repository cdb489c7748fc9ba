export ZK_SPLIT_V=0
for cfg in "X=1" "ZK_MSM_C_G2=12" "ZK_MSM_C_G2=11" "ZK_MSM_C_G2=10" "ZK_MSM_C=12" "ZK_MSM_C=11" "ZK_MSM_C=12 ZK_MSM_C_G2=11" "ZK_MSM_C=14"; do
  env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-primitives --no-tables 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('table-free $cfg', d['value'], d['ms_per_step'])"
done
