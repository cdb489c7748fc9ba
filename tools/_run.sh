python -m pytest tests/test_gpu_groth16.py tests/test_gpu_configs.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -4
for cfg in "ZK_SHARE_SORT=1" "ZK_SHARE_SORT=0"; do
  env $cfg python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-primitives 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'])"
  env $cfg python bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-primitives --no-tables 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('table-free $cfg', d['value'], d['ms_per_step'])"
done
