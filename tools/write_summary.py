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
notes = open(sys.argv[4]).read() if len(sys.argv) > 4 else ""
txt = """# %(R)s -- rocprofv3 summaries of the final round-%(rnd)d build (un-instrumented product library, MI355X, gpurun boxes)

Produced by `tools/profile_round.sh %(tag)s <workloads>` (each command with `python3` directly after `--`), digested by
`tools/profile_digest.py gpurun_out/%(tag)s %(rnd)d` and laid out by `tools/write_summary.py` (round-specific notes: last section).  Raw
statistics: `%(R)s_kernel_stats_{warm,cold}_<W>.csv`; the bench lines: `%(R)s_bench_<W>.json` (plain runs: warm headline + `roofline.cold`),
`%(R)s_bench_under_rocprofv3_{warm,cold}_<W>.json` (what the traced commands printed).

## States

* **warm** = `bench.py --state warm`: back-to-back launches over ONE input set, which never leaves the 256 MiB Infinity Cache: the fraction
  of the HBM roofline at cache-resident inputs (what a closed loop over one handle sees).
* **cold** = `bench.py --state cold`: every launch of the run rotates over 24-42 independent input sets (own handle, q and output
  buffers; >= 640 MiB touched between two uses of a set): inputs come from HBM.  The plain `bench.py` line carries both
  (`roofline.state`, `roofline.cold`).

## Kernel trace: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload W --no-cpu-baseline --rollout 0 --host-path 0 --secondary 0 --launch graph --state {warm,cold}`

(12 020+ dispatches each: warm-up + 2 x 30 repetitions x 200 launches.  Algorithmic bytes per launch, SURVEY 8d: C2 512 B x 4 096,
C3 / C3N 384 B x 65 536, C5 696 B x 65 536, C3F 600 B x 65 536, C5F 996 B x 65 536.  frac = algorithmic bytes / mean duration / 8 TB/s.)

%(table)s

Kernel names: `cycle_kernel_s` = a lean variant entered through its ten scalar arguments (preloaded into SGPRs), `cycle_kernel_x` = any
other variant (the same scalars in front of its argument block), `cycle_kernel_m` = the variants for differing decay orders; template
parameters <io type, joints, nullspace module, PLAIN, rollout, straight-line field path, LEAN (1 lean, 3 publishing lean), compile-time
flags, persistent, aux block, waves per SIMD, uniform repeller image, [order planes,] DH pattern>; of `cycle_sub8_kernel[_x]`: <io type,
joints, nullspace module, DH pattern>.

**How the traces were taken.**  The traced commands replay their launches from a hipGraph (`--launch graph`, what the plain command's `auto`
mode picks at 200 launches per region): back to back under the tracer as in the untraced run.  Launched one by one, the tracer's
per-dispatch work makes the process host-bound and the recorded durations of the short kernels bimodal and inflated (round 3:
`r03_trace_gaps.txt`); those one-by-one traces are kept as `%(R)s_kernel_stats_direct_{warm,cold}_<W>.csv`.  The traced means run 1-5 %%
above the untraced launch period of the last column (every dispatch carries the tracer's completion signal and timestamps; a kernel shorter
than ~5 us is held at the tracer's own dispatch period -- the round's notes): `bench.py`
prints the untraced HIP-event figure as `roofline.frac`, which is what its contract defines, and the traced means beside it
(`roofline.kernel_trace`, from `profiles/kernel_trace.json`).

## HBM traffic in the cold state: one counter per pass, `rocprofv3 --pmc FETCH_SIZE --kernel-trace ...` / `--pmc WRITE_SIZE ...` (`--state cold --steps 40 --reps 2`; median over the dispatches)

%(traffic)s

(FETCH_SIZE doubled: gfx950 counts 64 B per 128-B request of a 16-B-per-lane stream; WRITE_SIZE exact -- the guide's HBM section.
`profiles/pmc_traffic.json` is what `bench.py` quotes as `roofline.traffic`.  The bench workloads read the uniform repeller image, 16
instead of 32 bytes a slot: below the algorithmic bytes.)

## SQ counters per wave (= per 64 arms; C2: per 8 arms), warm state, `rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace ...`

%(sq)s

(Units: quad-cycles; instruction counts are exact, wait cycles are inflated by the profiler.  `profiles/pmc_sq.json` is what `bench.py`
quotes as `roofline.valu`: VALU instructions of a wave x 4 cycles / 2.2 GHz = the time the SIMD needs to issue them.)

%(notes)s
""" % dict(R=R, rnd=rnd, tag=tag, table=table, traffic=traffic, sq=sq, notes=notes)
open(os.path.join(ROOT, "profiles", "%s_summary.md" % R), "w").write(txt)
print("wrote profiles/%s_summary.md" % R)
