#!/bin/bash
# Profiles of one round, on the GPU box:  tools/profile_round.sh <tag> [workloads...]   (writes gpurun_out/<tag>/...)
# Per workload: kernel-trace statistics of the bench command in the WARM state (back-to-back launches over one input set,
# Infinity-Cache resident) and in the COLD state (`--state cold`: every launch rotates over >= 24 input sets, inputs from HBM)
# -- the average kernel durations the roofline lines must agree with.  The traced commands replay their launches from a hipGraph
# (`--launch graph`, what the plain command's `auto` picks at 200 steps per region): launched one by one, the tracer's own work per
# dispatch makes the process host-bound (8-11 us per launch) and the recorded durations of short kernels bimodal and inflated (they
# include queueing: profiles/r03_summary.md); one such trace is kept per workload as `direct_trace_<W>` for the record.
# Then SEPARATE counter passes (never combined with
# tracing other than --kernel-trace): HBM traffic (FETCH_SIZE, WRITE_SIZE; cold state, one counter per pass) and SQ
# instruction / wait counters.  The program follows `--` directly (python3).
set -u
tag=${1:-prof}; shift || true
wls=${@:-C3 C3N C5 C3F C5F C2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
common="--no-cpu-baseline --rollout 0 --host-path 0 --secondary 0"
short="--steps 40 --warmup 5 --reps 2"
for w in $wls; do
  echo "bench $w"; timeout -k 10 300 python3 $R/bench.py --workload $w $common --launch direct > $out/bench_$w.json 2> $out/bench_$w.err || echo "bench $w failed"
  echo "warm trace $w"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$w -- python3 $R/bench.py --workload $w $common --launch graph --state warm > $out/bench_under_rocprof_$w.json 2> $out/trace_$w.err || echo "trace $w failed"
  echo "cold trace $w"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cold_trace_$w -- python3 $R/bench.py --workload $w $common --launch graph --state cold > $out/bench_cold_under_rocprof_$w.json 2> $out/cold_trace_$w.err || echo "cold trace $w failed"
  echo "direct trace $w"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/direct_trace_$w -- python3 $R/bench.py --workload $w $common --launch direct --state warm > $out/bench_direct_under_rocprof_$w.json 2> $out/direct_trace_$w.err || echo "direct trace $w failed"
  for c in FETCH_SIZE WRITE_SIZE; do   # one memory counter per pass: together they exceed what the hardware collects at once
    echo "pmc $c $w"; timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_mem_${w}_$c -- python3 $R/bench.py --workload $w $common --launch direct --state cold $short > /dev/null 2> $out/pmc_mem_${w}_$c.err || echo "pmc $c $w failed"
  done
  echo "pmc sq $w"; timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $out/pmc_sq_$w -- python3 $R/bench.py --workload $w $common --launch direct --state warm --steps 20 --warmup 5 --reps 3 > /dev/null 2> $out/pmc_sq_$w.err || echo "pmc sq $w failed"
done
ls $out
