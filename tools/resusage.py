#!/usr/bin/env python
"""Registers, scratch and LDS of every kernel variant of the last build (vfclik_amd/csrc/nj*_kernels.resusage.txt,
written by -Rpass-analysis=kernel-resource-usage):  python tools/resusage.py [--scratch-only]"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for f in sorted(glob.glob(os.path.join(ROOT, "vfclik_amd", "csrc", "nj*_kernels.resusage.txt"))):
    txt = open(f).read()
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", txt, re.S):
        rows.append(m.groups())
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
print("template parameters of cycle_kernel: <io type, joints, nullspace module, PLAIN, rollout, straight-line field path, LEAN, compile-time flags>")
for r, n in zip(rows, names):
    n = n.replace("void vfik::(anonymous namespace)::", "").replace("(vfik::KArgs)", "")
    if "--scratch-only" in sys.argv and r[3] == "0":
        continue
    print("%-70s VGPR %3s AGPR %3s scratch %4s B/lane" % (n, r[1], r[2], r[3]))
