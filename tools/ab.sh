#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab.sh <other.so> <rounds> <bench.py args...>
# prints value / table-free / batched figures of every run (A = the in-tree build, B = the other one)
OTHER=$1; N=$2; shift 2
for i in $(seq 1 $N); do
  for v in A B; do
    if [ $v = B ]; then RUN="python tools/ab_run.py $OTHER bench.py"; else RUN="python bench.py"; fi
    $RUN "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
o={'v':'$v','value':d.get('value'),'ms':d.get('ms_per_step')}
if 'table_free' in d: o['tf']=d['table_free']['value']
if 'pipelined' in d: o['pipe']=d['pipelined']['proofs_per_s']
if 'batched' in d and isinstance(d['batched'],list): o['batched']=[b['proofs_per_s'] for b in d['batched']]
if 'primitives' in d: o['prim']={k[:12]:v['ms'] for k,v in d['primitives'].items()}
print(json.dumps(o))"
  done
done
