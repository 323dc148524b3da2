#!/bin/bash
# Profiles of one round, on the GPU box:  tools/profile_round.sh <tag>   (writes gpurun_out/<tag>/...)
# Kernel-trace statistics of the plain bench command per workload (the average kernel duration the roofline line must
# agree with), then SEPARATE counter passes (never combined with tracing other than --kernel-trace): HBM traffic
# (FETCH_SIZE, WRITE_SIZE) and SQ instruction / wait counters.  The program follows `--` directly (python3).
set -u
tag=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
common="--no-cpu-baseline --rollout 0 --host-path 0"
for w in C3 C3N C5 C2; do
  echo "trace $w"; timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$w -- python3 $R/bench.py --workload $w $common > $out/bench_under_rocprof_$w.json 2> $out/trace_$w.err || echo "trace $w failed"
done
short="--steps 20 --warmup 5 --reps 3"
for w in C3 C3N C5; do
  for c in FETCH_SIZE WRITE_SIZE; do   # one memory counter per pass: together they exceed what the hardware collects at once
    echo "pmc $c $w"; timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_mem_${w}_$c -- python3 $R/bench.py --workload $w $common $short > /dev/null 2> $out/pmc_mem_${w}_$c.err || echo "pmc $c $w failed"
  done
  echo "pmc sq $w"; timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $out/pmc_sq_$w -- python3 $R/bench.py --workload $w $common $short > /dev/null 2> $out/pmc_sq_$w.err || echo "pmc sq $w failed"
done
# the un-profiled lines of the same build, for the record
for w in C3 C3N C5 C2; do echo "bench $w"; timeout -k 10 180 python3 $R/bench.py --workload $w $common > $out/bench_$w.json 2> $out/bench_$w.err; done
ls $out
