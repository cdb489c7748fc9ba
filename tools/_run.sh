python bench.py --no-cpu-baseline --no-primitives > gpurun_out/b_c4.json 2> gpurun_out/b_c4.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/b_c4.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v for k,v in d.items() if k in('table_free','pipelined')})
PY
python tools/c5_bls381.py 22 > gpurun_out/c5_22_new3.json 2> gpurun_out/c5_22_new3.err; tail -2 gpurun_out/c5_22_new3.err
python -c "
import json; d=json.load(open('gpurun_out/c5_22_new3.json')); print(d['prove_s'], d['all_parties_equal'], d['distributed_equals_local'])"
