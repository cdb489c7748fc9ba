# Regenerates the round's evidence on a GPU box (gpurun -- 'bash tools/refresh_profiles.sh'); results land in gpurun_out/
# as r04_*; copy what is to be judged into profiles/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python bench.py > $O/r04_bench_c4.json 2> $O/r04_bench_c4.err
python bench.py --workload c2 > $O/r04_bench_c2.json 2> $O/r04_bench_c2.err
python bench.py --workload c3 > $O/r04_bench_c3.json 2> $O/r04_bench_c3.err
cd /tmp && export TMPDIR=/tmp
for v in c4 c4tf c2 c3 b8 b8x2 b8wide; do
  P=bench.py
  case $v in
    c4) A="--no-cpu-baseline --no-primitives --steps 20 --warmup 5";;
    c4tf) A="--no-cpu-baseline --no-primitives --no-tables --steps 20 --warmup 5";;
    c2) A="--workload c2 --no-cpu-baseline --steps 10 --warmup 3";;
    c3) A="--workload c3 --no-cpu-baseline --no-tables --steps 5 --warmup 2";;
    b8) P=tools/batch_probe.py; A="8 --reps 4";;
    b8x2) P=tools/batch_probe.py; A="8 --reps 6 --inflight 2";;
    b8wide) P=tools/batch_probe.py; A="8 --reps 4"; export ZK_QUAD_THREADS=256; export ZK_BATCH_ORDER=-;;
  esac
  rm -rf $O/prof_$v $O/pmcf_$v $O/pmcw_$v
  rocprofv3 --kernel-trace -d $O/prof_$v -o p -- python3 $R/$P $A > $O/prof_$v.log 2>&1
  unset ZK_QUAD_THREADS ZK_BATCH_ORDER
  if [ $v = c4 ] || [ $v = c2 ] || [ $v = c3 ] || [ $v = b8 ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf_$v -o p -- python3 $R/$P $A > $O/pmcf_$v.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw_$v -o p -- python3 $R/$P $A > $O/pmcw_$v.log 2>&1
  fi
done
rm -rf $O/pmc_sq $O/pmc_u $O/pmc_sq_b8
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -o p -- python3 $R/bench.py --workload c3 --no-cpu-baseline --no-tables --steps 5 --warmup 2 > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_u -o p -- python3 $R/bench.py --no-cpu-baseline --no-primitives --steps 10 --warmup 3 > $O/pmc_u.log 2>&1
rm -rf $O/pmc_c2a $O/pmc_c2b
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_c2a -o p -- python3 $R/bench.py --workload c2 --no-cpu-baseline --steps 5 --warmup 2 > $O/pmc_c2a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/pmc_c2b -o p -- python3 $R/bench.py --workload c2 --no-cpu-baseline --steps 5 --warmup 2 > $O/pmc_c2b.log 2>&1
cd $R
python tools/sq_c2_summary.py $O/pmc_c2a $O/pmc_c2b $O/r04_c2_sq_counters.json > /dev/null
for v in c4 c4tf c2 c3 b8 b8x2 b8wide; do
  f=$(find $O/prof_$v -name "p_results.db" | head -1)
  python tools/kernel_stats.py $f > $O/r04_${v}_kernel_stats.csv
  case $v in
    c4|c4tf) python tools/timeline.py $f 3 > $O/r04_${v}_timeline.txt;;
    b8|b8wide) python tools/timeline.py $f 1 300 > $O/r04_${v}_timeline.txt;;
    b8x2) python tools/timeline.py $f 1 3000 > $O/r04_${v}_timeline.txt;;
  esac
  rm -f $f
  if [ -d $O/pmcf_$v ]; then python tools/pmc_summary.py $O/pmcf_$v $O/pmcw_$v $O/r04_${v}_pmc_hbm.json "$v, round 4" > /dev/null; rm -rf $O/pmcf_$v $O/pmcw_$v; fi
done
python tools/sq_summary.py $O/pmc_sq $O/r04_c3_kernel_stats.csv $O/r04_c3_sq_counters.json > /dev/null
python tools/lane_util.py $O/pmc_u $O/r04_c4_lane_utilisation.json
python tools/acc_batch_solo.py 8 2>&1 | grep -v amdgpu > $O/r04_acc_batch_solo.txt
python tools/acc_batch_solo.py 1 2>&1 | grep -v amdgpu >> $O/r04_acc_batch_solo.txt
python tools/c5_bls381.py 24 > $O/r04_c5.json 2> $O/r04_c5.err
cd /tmp; rm -rf $O/prof_c5
rocprofv3 --kernel-trace -d $O/prof_c5 -o p -- python3 $R/tools/c5_bls381.py 24 > $O/prof_c5.log 2>&1
cd $R
f=$(find $O/prof_c5 -name "p_results.db" | head -1); python tools/kernel_stats.py $f > $O/r04_c5_kernel_stats.csv; rm -rf $O/prof_c5
python bench.py --workload c5 --steps 3 --warmup 1 > $O/r04_bench_c5.json 2> $O/r04_bench_c5.err
python tools/rank_latency.py 100 2>/dev/null | tail -1 > $O/r04_rank_latency.json
