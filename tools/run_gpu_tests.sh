#!/bin/bash
# Run every -m gpu test in its own process with a hard timeout, so a hung kernel costs seconds, not the session.
# usage: tools/run_gpu_tests.sh [per-test-timeout-seconds] [pytest -k expression]
T=${1:-120}
K=${2:-}
mkdir -p gpurun_out
ids=$(python -m pytest tests -m gpu --collect-only -q ${K:+-k "$K"} 2>/dev/null | grep '::')
pass=0; fail=0
: > gpurun_out/gpu_tests.log
for id in $ids; do
  out=$(timeout $T python -m pytest "$id" -x -q --no-header -p no:cacheprovider 2>&1)
  rc=$?
  if [ $rc -eq 0 ]; then pass=$((pass+1)); echo "PASS $id" >> gpurun_out/gpu_tests.log
  else fail=$((fail+1)); echo "FAIL($rc) $id" >> gpurun_out/gpu_tests.log; echo "$out" | tail -25 >> gpurun_out/gpu_tests.log; fi
done
echo "passed=$pass failed=$fail" | tee -a gpurun_out/gpu_tests.log
