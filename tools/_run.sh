set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
mkdir -p $O
python bench.py > $O/bench_c4.json 2> $O/bench_c4.err; tail -2 $O/bench_c4.err
python bench.py --workload c2 --steps 30 --warmup 3 > $O/bench_c2.json 2>> $O/bench_c4.err
python bench.py --workload c3 --steps 5 --warmup 1 > $O/bench_c3.json 2>> $O/bench_c4.err
# kernel trace + stats of the headline loop
rocprofv3 --kernel-trace -d $O/kt_c4 -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-primitives > $O/kt_c4.log 2>&1
python tools/kernel_stats.py $O/kt_c4/p_results.db > $O/c4_kernel_stats.csv; python tools/timeline.py $O/kt_c4/p_results.db 2 > $O/c4_timeline.txt
rocprofv3 --kernel-trace -d $O/kt_c4tf -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-primitives --no-tables > $O/kt_c4tf.log 2>&1
python tools/kernel_stats.py $O/kt_c4tf/p_results.db > $O/c4_table_free_kernel_stats.csv; python tools/timeline.py $O/kt_c4tf/p_results.db 2 > $O/c4_table_free_timeline.txt
rocprofv3 --kernel-trace -d $O/kt_c2 -o p -- python3 bench.py --workload c2 --steps 20 --warmup 3 > $O/kt_c2.log 2>&1
python tools/kernel_stats.py $O/kt_c2/p_results.db > $O/c2_kernel_stats.csv
rocprofv3 --kernel-trace -d $O/kt_c3 -o p -- python3 bench.py --workload c3 --steps 3 --warmup 1 > $O/kt_c3.log 2>&1
python tools/kernel_stats.py $O/kt_c3/p_results.db > $O/c3_kernel_stats.csv
# PMC passes (separate runs, counters only)
for w in c4 c2 c3; do
  if [ $w = c4 ]; then A="--steps 5 --warmup 1 --no-cpu-baseline --no-primitives"; elif [ $w = c2 ]; then A="--workload c2 --steps 5 --warmup 1"; else A="--workload c3 --steps 2 --warmup 1"; fi
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f_$w -o p -- python3 bench.py $A > $O/pmc_f_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w_$w -o p -- python3 bench.py $A > $O/pmc_w_$w.log 2>&1
  python tools/pmc_summary.py $O/pmc_f_$w $O/pmc_w_$w $O/${w}_pmc_hbm.json "bench.py $A" > /dev/null
done
ls -la $O | head -40
rm -rf $O/pmc_f_* $O/pmc_w_* $O/kt_*/
