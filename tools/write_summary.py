#!/usr/bin/env python
"""profiles/r<NN>_summary.md from the digest of a profile round:
    python tools/profile_digest.py gpurun_out/<tag> <round> > /tmp/digest.txt && python tools/write_summary.py /tmp/digest.txt <round> <tag>
The three blocks of the digest (kernel trace table, HBM traffic, SQ counters) are embedded verbatim; the prose around them is the
round's reading guide."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dig = open(sys.argv[1]).read()
rnd, tag = int(sys.argv[2]), sys.argv[3]
parts = [p for p in dig.split("\n\n") if p.strip()]
table, traffic, sq = parts[0], parts[1], parts[2]
R = "r%02d" % rnd
txt = """# %(R)s -- rocprofv3 summaries of the final round-%(rnd)d build (un-instrumented product library, MI355X, gpurun boxes)

Produced by `tools/profile_round.sh %(tag)s <workloads>` (two calls: C3 C3N C5, then C3F C5F C2 C2F; each command with `python3` directly after
`--`), digested by `tools/profile_digest.py gpurun_out/%(tag)s %(rnd)d` and laid out by `tools/write_summary.py`.  Raw statistics:
`%(R)s_kernel_stats_{warm,cold}_<W>.csv`; the bench lines: `%(R)s_bench_<W>.json` (plain runs: warm headline + `roofline.cold`),
`%(R)s_bench_under_rocprofv3_{warm,cold}_<W>.json` (what the traced commands printed), `%(R)s_bench_driver_command_steps20.json` (the driver's
command), `%(R)s_bench_n2_gloo_single_device.json`.

## States

* **warm** = `bench.py --state warm`: back-to-back launches over ONE input set.  A C3 launch reads 20.7 MB; re-read by the next launch it
  never leaves the 256 MiB Infinity Cache.  Rounds 1 and 2 measured only this state and called the bound "hbm"; it is the fraction of
  the HBM roofline at cache-resident inputs (and what a closed loop over one handle sees).
* **cold** = `bench.py --state cold`: every launch of the run rotates over 24-34 independent input sets (own handle, q and output
  buffers; >= 640 MiB touched between two uses of a set): inputs come from HBM.  The plain `bench.py` line carries both
  (`roofline.state`, `roofline.cold`).

## Kernel trace: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload W --no-cpu-baseline --rollout 0 --host-path 0 --launch graph --state {warm,cold}`

(12 020+ dispatches each: warm-up + 2 x 30 repetitions x 200 launches.  Algorithmic bytes per launch, SURVEY 8d: C2 512 B x 4 096,
C3 / C3N 384 B x 65 536, C5 696 B x 65 536, C3F 600 B x 65 536, C5F 996 B x 65 536.  frac = algorithmic bytes / mean duration / 8 TB/s.)

%(table)s

Kernel names: `cycle_kernel_s` = a lean variant entered through its ten scalar arguments (preloaded into SGPRs), `cycle_kernel_x` = any
other variant (the same scalars in front of its argument block); template parameters <io type, joints, nullspace module, PLAIN, rollout,
straight-line field path, LEAN (1 lean, 3 publishing lean), compile-time flags, persistent, aux block, waves per SIMD>; of
`cycle_sub8_kernel[_x]`: <io type, joints, nullspace module>.

**How the traces were taken, and traced against untraced.**  The traced commands replay their launches from a hipGraph (`--launch graph`,
what the plain command's `auto` mode picks at 200 launches per region): back to back under the tracer as in the untraced run, start-to-start
= duration (`%(R)s_trace_gaps.txt`).  Launched one by one, the tracer's per-dispatch work makes the process host-bound (8-11 us per launch for
the short kernels: column "the traced process's own HIP events" of the `direct` traces), and the recorded durations turn bimodal and inflated --
C3: 6 216 ns mean (median 6 160, p10 4 640, p90 7 720) against 5 164 (5 040 / 4 960 / 5 280) graph-replayed on the same box, and a kernel that
follows an idle gap of more than 3 us runs 0.7-1.6 us longer in every trace (`tools/trace_gaps.py`).  Those one-by-one traces are kept as
`%(R)s_kernel_stats_direct_{warm,cold}_<W>.csv` for C3, C3N, C5 (for the launches of 10 us and more the two kinds agree: C3F 10 324 / 10 309).
The traced means run 1-5 %% above the untraced launch period of the last column (every dispatch carries the tracer's completion signal
and timestamps): `bench.py` prints the untraced HIP-event figure as `roofline.frac`, which is what its contract defines, and the traced
means beside it (`roofline.kernel_trace`, from `profiles/kernel_trace.json`).
Round 2 (warm only, launched one by one): C3 5 674 ns (0.554), C3N 7 565 (0.416), C5 10 826 (0.527), C2 5 153 (0.051).

## HBM traffic in the cold state: one counter per pass, `rocprofv3 --pmc FETCH_SIZE --kernel-trace ...` / `--pmc WRITE_SIZE ...` (`--state cold --steps 40 --reps 2`; median over the dispatches)

%(traffic)s

(FETCH_SIZE doubled: gfx950 counts 64 B per 128-B request of a 16-B-per-lane stream; WRITE_SIZE exact -- the guide's HBM section.  The
same figures as round 2's warm passes: FETCH_SIZE counts Infinity-Cache hits too.  `profiles/pmc_traffic.json` is what `bench.py` quotes as
`roofline.traffic`.)  C3 / C5 read the compact repeller image (24 instead of 32 bytes a slot): below the algorithmic bytes.  C3F writes
18.09 MB = 65 536 x (61 scalars + 8 of nullspace state) x 4 B; C5F 23.33 MB.

## SQ counters per wave (= per 64 arms; C2: per 8 arms), warm state, `rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace ...`

%(sq)s

(Units: quad-cycles; instruction counts are exact, wait cycles are inflated by the profiler.)  Round 2: C3 VALU 1 268 / active 1 608,
C3N 2 013 / 2 457, C5 3 457 / 4 049.  C2's and C2F's kernels are the eight-lanes-per-arm ones: 944 / 1 810 VALU instructions per wave of 8 arms.

## Other artefacts of the round

* `%(R)s_bench_driver_command_steps20.json` -- `python3 bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command).
* `%(R)s_bench_n2_gloo_single_device.json` -- `python3 bench.py --gpus 2 --single-device --dist-backend gloo --steps 20 --warmup 5 --gather`:
  the parent spawned both ranks (each pinned to its own CPUs), two ranks SHARING one GPU, `ShardedEngine.gather` collated 131 072 rows.
  (8-GPU scaling is the driver's to measure.)
* Small batches: `%(R)s_latency_small_f64_4obst.txt`, `%(R)s_latency_small_f32_8obst.txt`; floor: `%(R)s_ubench_launch.txt` (+ `_host_kernarg`);
  cross-lane prices: `%(R)s_ubench_xlane.txt`.  Beyond one wave per SIMD: `%(R)s_batch_scaling.txt` (rounds / two waves per SIMD / persistent),
  `%(R)s_stamps_131072_arms_in_rounds.txt`.  Field paths: `%(R)s_general_path.txt`.
* Drop-in path: `%(R)s_ccb_rate.txt` (ControlCycleBatch.cycle() with ports / step_arrays with every output / qdot_out only).
* Stamps (diagnostic build): `%(R)s_stamps_C3_{warm,cold}.txt`, `%(R)s_stamps_C3N_{warm,cold}.txt`, `%(R)s_stamps_C3F_warm.txt`.
* A/B log: `%(R)s_ab_experiments.md` with its raw files (`%(R)s_ab_*.txt`): preloaded scalar kernel arguments, uniform repeller image, two waves per
  SIMD, aux block, and the rejected ones (nt output stores, block size, q first, SGPR-base requests, iterative-ilp scheduling, ...).
* Independent batches in flight on one GPU: `%(R)s_two_handles.txt`; host-pointer calls without a copy: `%(R)s_ccb_rate_zero_copy_ab.txt`;
  field paths: `%(R)s_general_path.txt`; traced durations against launch cadence: `%(R)s_trace_gaps.txt`.
""" % dict(R=R, rnd=rnd, tag=tag, table=table, traffic=traffic, sq=sq)
open(os.path.join(ROOT, "profiles", "%s_summary.md" % R), "w").write(txt)
print("wrote profiles/%s_summary.md" % R)
