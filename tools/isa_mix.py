#!/usr/bin/env python
"""Instruction mix of one kernel of libvfik_hip.so (static count over the disassembly; straight-line kernels, so it is
close to the dynamic count).  python tools/isa_mix.py 'cycle_kernel<float, 14, true, true, false, true, 1>'"""
import collections
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def main():
    want = sys.argv[1]
    lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "vfclik_amd", "csrc", "libvfik_hip.so")
    with tempfile.TemporaryDirectory() as d:
        subprocess.run(["cp", lib, os.path.join(d, "lib.so")], check=True)
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=d, check=True, capture_output=True)
        for co in sorted(glob.glob(os.path.join(d, "lib.so.*gfx950"))):
            txt = subprocess.run([OBJDUMP, "-d", "-C", co], capture_output=True, text=True).stdout
            blocks = re.split(r"\n(?=[0-9a-f]{16} <)", txt)
            for b in blocks:
                head = b.split("\n", 1)[0]
                if want not in head:
                    continue
                ops = collections.Counter()
                for line in b.split("\n")[1:]:
                    m = re.match(r"\s+([a-z_0-9]+)", line)
                    if m:
                        ops[m.group(1)] += 1
                groups = collections.Counter()
                for op, c in ops.items():
                    if op.startswith("v_accvgpr"): g = "accvgpr moves"
                    elif op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64")): g = "f64 fma/mul/add"
                    elif op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")): g = "f64 transcendental"
                    elif op.startswith(("v_mov", "v_cndmask", "v_cmp", "v_readlane", "v_writelane", "v_readfirstlane")): g = "moves/selects/compares"
                    elif op.startswith("v_cvt"): g = "conversions"
                    elif op.startswith("v_"): g = "other VALU"
                    elif op.startswith("s_waitcnt"): g = "s_waitcnt"
                    elif op.startswith("s_load"): g = "SMEM"
                    elif op.startswith("s_"): g = "SALU/other scalar"
                    elif op.startswith("ds_"): g = "LDS"
                    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): g = "VMEM"
                    else: g = "other"
                    groups[g] += c
                print(head.strip())
                tot = sum(ops.values())
                for g, c in groups.most_common():
                    print("  %-26s %5d" % (g, c))
                print("  %-26s %5d" % ("total", tot))
                print("  top:", ", ".join("%s %d" % kv for kv in ops.most_common(14)))
                return
    print("kernel not found")


if __name__ == "__main__":
    main()
