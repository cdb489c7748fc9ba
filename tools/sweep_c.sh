#!/bin/bash
# Window-size sweeps on one box: C4 with fixed-base tables (G1 / G2 table bits) and C3 table-free (msm_c).
# usage (GPU box): bash tools/sweep_c.sh [c4|c3 ...]   -> one line per setting
run() { "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$TAG', d['value'], d['ms_per_step'], [b['proofs_per_s'] for b in d.get('batched', [])] if isinstance(d.get('batched'), list) else '')"; }
PARTS=${*:-c4 c3}
case " $PARTS " in *" c4 "*)
  for rep in 1 2; do
    for c in 13 14 15 16; do TAG="c4 msm_table_c(g1)=$c g2=15"; ZK_BENCH_OPTIONS="msm_table_c=$c,msm_table_c_g2=15" run python bench.py --no-cpu-baseline --no-primitives; done
  done
  for c in 14 15 16; do TAG="c4 g1=15 msm_table_c_g2=$c"; ZK_BENCH_OPTIONS="msm_table_c=15,msm_table_c_g2=$c" run python bench.py --no-cpu-baseline --no-primitives; done
  for c in 15 16; do TAG="c4 full msm_table_c(g1)=$c g2=15"; ZK_BENCH_OPTIONS="msm_table_c=$c,msm_table_c_g2=15" run python bench.py --no-cpu-baseline; done;;
esac
case " $PARTS " in *" c3 "*)
  for c in 0 14 15 16 17 18; do TAG="c3 msm_c=$c"; ZK_BENCH_OPTIONS="msm_c=$c" run python bench.py --workload c3 --no-cpu-baseline --no-tables; done;;
esac
