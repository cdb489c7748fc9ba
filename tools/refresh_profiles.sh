# Regenerates the round's evidence on a GPU box (gpurun -- 'bash tools/refresh_profiles.sh'); results land in gpurun_out/
# as r02_*; copy what is to be judged into profiles/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python bench.py > $O/r02_bench_c4.json 2> $O/r02_bench_c4.err
python bench.py --workload c2 > $O/r02_bench_c2.json 2> $O/r02_bench_c2.err
python bench.py --workload c3 > $O/r02_bench_c3.json 2> $O/r02_bench_c3.err
cd /tmp && export TMPDIR=/tmp
for v in c4 c4tf c2 c3; do
  case $v in
    c4) A="--no-cpu-baseline --no-primitives --steps 20 --warmup 5";;
    c4tf) A="--no-cpu-baseline --no-primitives --no-tables --steps 20 --warmup 5";;
    c2) A="--workload c2 --no-cpu-baseline --steps 10 --warmup 3";;
    c3) A="--workload c3 --no-cpu-baseline --steps 5 --warmup 2";;
  esac
  rm -rf $O/prof_$v $O/pmcf_$v $O/pmcw_$v
  rocprofv3 --kernel-trace -d $O/prof_$v -o p -- python3 $R/bench.py $A > $O/prof_$v.log 2>&1
  if [ $v != c4tf ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf_$v -o p -- python3 $R/bench.py $A > $O/pmcf_$v.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw_$v -o p -- python3 $R/bench.py $A > $O/pmcw_$v.log 2>&1
  fi
done
rm -rf $O/pmc_sq $O/pmc_u
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -o p -- python3 $R/bench.py --workload c3 --no-cpu-baseline --steps 5 --warmup 2 > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_u -o p -- python3 $R/bench.py --no-cpu-baseline --no-primitives --steps 10 --warmup 3 > $O/pmc_u.log 2>&1
cd $R
for v in c4 c4tf c2 c3; do
  f=$(find $O/prof_$v -name "p_results.db" | head -1)
  python tools/kernel_stats.py $f > $O/r02_${v}_kernel_stats.csv
  if [ $v = c4 ] || [ $v = c4tf ]; then python tools/timeline.py $f 3 > $O/r02_${v}_timeline.txt; fi
  rm -f $f
  if [ $v != c4tf ]; then python tools/pmc_summary.py $O/pmcf_$v $O/pmcw_$v $O/r02_${v}_pmc_hbm.json "bench.py $v, round 2 final" > /dev/null; rm -rf $O/pmcf_$v $O/pmcw_$v; fi
done
python tools/sq_summary.py $O/pmc_sq $O/r02_c3_kernel_stats.csv $O/r02_c3_sq_counters.json > /dev/null
python tools/c5_bls381.py 24 > $O/r02_c5.json 2> $O/r02_c5.err
python bench.py --workload c5 --steps 3 --warmup 1 > $O/r02_bench_c5.json 2> $O/r02_bench_c5.err
python tools/rank_latency.py 100 2>/dev/null | tail -1 > $O/r02_rank_latency.json
python tools/lane_util.py $O/pmc_u $O/r02_c4_lane_utilisation.json
