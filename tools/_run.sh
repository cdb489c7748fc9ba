for q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-primitives 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('hwq $q', d['value'], d['ms_per_step'])"
done
ZK_HOST_THREADS=32 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-primitives 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('threads32', d['value'], d['ms_per_step'])"
ZK_MSM_SEG=8 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-primitives 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('seg8', d['value'], d['ms_per_step'])"
ZK_MSM_SEG=32 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-primitives 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('seg32', d['value'], d['ms_per_step'])"
for c in 14 15 17; do
ZK_TABLE_C=$c python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-primitives 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('table_c $c', d['value'], d['ms_per_step'])"
done
