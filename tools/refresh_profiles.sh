# Regenerates the round's evidence on a GPU box (gpurun -- 'bash tools/refresh_profiles.sh [parts]'); results land in
# gpurun_out/ as r06_*; copy what is to be judged into profiles/.  parts (default: all): calib bench c4 c2 c3 dpp dealer c5 misc
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
PARTS=${*:-calib bench c4 c2 c3 dpp dealer c5 misc}
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
SQ="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES"
cd $R
if has calib; then
  # counter calibration first: tools/pmc_summary.py corrects every kernel by the factor of its access shape
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/calf $O/calw
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calf -o p -- $R/tools/pmc_calib > $O/calf.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/calw -o p -- $R/tools/pmc_calib > $O/calw.log 2>&1
  cd $R
  python tools/pmc_calibrate.py $O/calf $O/calw $O/r06_pmc_calibration.json > /dev/null && cp $O/r06_pmc_calibration.json profiles/
  rm -rf $O/calf $O/calw
fi
if has bench; then
  python bench.py > $O/r06_bench_c4.json 2> $O/r06_bench_c4.err
  python bench.py --workload c2 > $O/r06_bench_c2.json 2> $O/r06_bench_c2.err
  python bench.py --workload c3 --no-tables > $O/r06_bench_c3.json 2> $O/r06_bench_c3.err     # table-free, as the c3 profile passes below
fi
cd /tmp && export TMPDIR=/tmp
for v in c4 c4tf c2 c3 dpp dealer c5; do
  has $v || { [ $v = c4tf ] && has c4; } || continue
  P=bench.py
  case $v in
    c4) A="--no-cpu-baseline --no-primitives --steps 20 --warmup 5";;
    c4tf) A="--no-cpu-baseline --no-primitives --no-tables --steps 20 --warmup 5";;
    c2) A="--workload c2 --no-cpu-baseline --steps 10 --warmup 3";;
    c3) A="--workload c3 --no-cpu-baseline --no-tables --steps 5 --warmup 2";;
    dpp) P=tools/dpp_bench.py; A="bn254 20 10 bls12_381 24 3";;
    dealer) P=tools/dealer_bench.py; A="--reps 3";;
    c5) A="--workload c5 --no-cpu-baseline --steps 1 --warmup 0";;
  esac
  rm -rf $O/prof_$v $O/pmcf_$v $O/pmcw_$v $O/pmcs_$v
  rocprofv3 --kernel-trace -d $O/prof_$v -o p -- python3 $R/$P $A > $O/prof_$v.log 2>&1
  if [ $v != c4tf ] && [ $v != dealer ]; then
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcf_$v -o p -- python3 $R/$P $A > $O/pmcf_$v.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmcw_$v -o p -- python3 $R/$P $A > $O/pmcw_$v.log 2>&1
  fi
  if [ $v = dpp ] || [ $v = c5 ] || [ $v = c3 ]; then
    rocprofv3 --pmc $SQ --output-format csv -d $O/pmcs_$v -o p -- python3 $R/$P $A > $O/pmcs_$v.log 2>&1
  fi
done
if has c4; then
  rm -rf $O/pmc_u
  rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/pmc_u -o p -- python3 $R/bench.py --no-cpu-baseline --no-primitives --steps 10 --warmup 3 > $O/pmc_u.log 2>&1
fi
if has c2; then
  rm -rf $O/pmc_c2a $O/pmc_c2b
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_c2a -o p -- python3 $R/bench.py --workload c2 --no-cpu-baseline --steps 5 --warmup 2 > $O/pmc_c2a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $O/pmc_c2b -o p -- python3 $R/bench.py --workload c2 --no-cpu-baseline --steps 5 --warmup 2 > $O/pmc_c2b.log 2>&1
fi
cd $R
has c2 && python tools/sq_c2_summary.py $O/pmc_c2a $O/pmc_c2b $O/r06_c2_sq_counters.json > /dev/null
for v in c4 c4tf c2 c3 dpp dealer c5; do
  [ -d $O/prof_$v ] || continue
  f=$(find $O/prof_$v -name "p_results.db" | head -1)
  python tools/kernel_stats.py $f > $O/r06_${v}_kernel_stats.csv
  case $v in
    c4|c4tf) python tools/timeline.py $f 3 > $O/r06_${v}_timeline.txt;;
    c5) python tools/timeline.py $f 1 100000 > $O/r06_${v}_timeline.txt;;
  esac
  rm -rf $O/prof_$v
  if [ -d $O/pmcf_$v ]; then python tools/pmc_summary.py $O/pmcf_$v $O/pmcw_$v $O/r06_${v}_pmc_hbm.json "$v, round 6" > /dev/null; rm -rf $O/pmcf_$v $O/pmcw_$v; fi
done
[ -d $O/pmcs_c3 ] && python tools/sq_summary.py $O/pmcs_c3 $O/r06_c3_kernel_stats.csv $O/r06_c3_sq_counters.json > /dev/null
[ -d $O/pmcs_dpp ] && python tools/sq_generic.py $O/pmcs_dpp $O/r06_dpp_sq_counters.json "rocprofv3 --pmc $SQ -- python3 tools/dpp_bench.py bn254 20 10 bls12_381 24 3" > /dev/null
[ -d $O/pmcs_c5 ] && python tools/sq_generic.py $O/pmcs_c5 $O/r06_c5_sq_counters.json "rocprofv3 --pmc $SQ -- python3 bench.py --workload c5 --no-cpu-baseline --steps 1 --warmup 0" > /dev/null
rm -rf $O/pmcs_c3 $O/pmcs_dpp $O/pmcs_c5
[ -d $O/pmc_u ] && python tools/lane_util.py $O/pmc_u $O/r06_c4_lane_utilisation.json
if has dealer; then
  python bench.py --workload dealer > $O/r06_dealer.json 2> $O/r06_dealer.err
fi
if has misc; then
  python tools/c5_cpu_growth.py > $O/r06_c5_cpu_growth.json 2> $O/r06_c5_cpu_growth.err && cp $O/r06_c5_cpu_growth.json profiles/
  python tools/dpp_bench.py 2>&1 | grep -v amdgpu > $O/r06_dpp_bench.json
  python tools/c5_bls381.py 24 > $O/r06_c5.json 2> $O/r06_c5.err
  python tools/rank_latency.py 100 2>/dev/null | tail -1 > $O/r06_rank_latency.json
fi
if has c5; then
  cp $O/r06_c5_pmc_hbm.json profiles/ 2>/dev/null      # bench.py's c5 roofline.traffic reads it
  python bench.py --workload c5 --steps 3 --warmup 1 > $O/r06_bench_c5.json 2> $O/r06_bench_c5.err
fi
