python -m pytest tests/test_gpu_msm.py tests/test_gpu_groth16.py -m gpu -x -q --deselect tests/test_gpu_groth16.py::test_two_ranks_sharing_the_gpu_give_the_single_rank_proof 2>&1 | tail -5
for mode in base gate; do
  if [ $mode = gate ]; then export ZK_GATE_ACC=1; fi
  python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_e_$mode.json 2> gpurun_out/r02_bench_e.err; tail -3 gpurun_out/r02_bench_e.err
  python -c "
import json
d=json.load(open('gpurun_out/r02_bench_e_$mode.json'))
print('$mode', d['value'], d['ms_per_step'], d['table_free']['value'], d['pipelined']['proofs_per_s'])
print(d['primitives']['d_msm_g1_8x2^20_bn254']['ms'])
"
done
unset ZK_GATE_ACC
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace -d gpurun_out/prof_t -o p -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-primitives > gpurun_out/prof_t.log 2>&1; python tools/timeline.py gpurun_out/prof_t/p_results.db 2
