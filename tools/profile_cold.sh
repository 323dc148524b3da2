#!/bin/bash
# Cold-state (HBM-sourced) profiles of one round, on the GPU box:  tools/profile_cold.sh <tag> [workloads...]
# `bench.py --state cold`: every launch of the run rotates over >= 24 independent input sets (> 640 MiB touched between two
# uses of a set), so the kernel-trace average is that of launches whose inputs come from HBM, not from the Infinity Cache.
# The program follows `--` directly (python3).  Counter passes are separate and carry --kernel-trace only.
set -u
tag=${1:-cold}; shift || true
wls=${@:-C3 C3N C5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
common="--no-cpu-baseline --rollout 0 --host-path 0 --launch direct"
for w in $wls; do
  echo "bench both $w"; timeout -k 10 300 python3 $R/bench.py --workload $w $common > $out/bench_both_$w.json 2> $out/bench_both_$w.err || echo "bench both $w failed"
  echo "cold trace $w"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cold_trace_$w -- python3 $R/bench.py --workload $w $common --state cold > $out/bench_cold_under_rocprof_$w.json 2> $out/cold_trace_$w.err || echo "cold trace $w failed"
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "cold pmc $c $w"; timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/cold_pmc_${w}_$c -- python3 $R/bench.py --workload $w $common --state cold --steps 40 --warmup 5 --reps 2 > /dev/null 2> $out/cold_pmc_${w}_$c.err || echo "cold pmc $c $w failed"
  done
done
ls $out
