#!/bin/bash
# Build libvfik_hip.so of a git revision out of tree (for tools/ab_compare.py): tools/build_rev.sh <rev> <out.so>
set -e
rev=$1; out=$(realpath -m "$2"); d=$(mktemp -d /tmp/vfik_rev.XXXX)
root=$(git rev-parse --show-toplevel)
mkdir -p $d/vfclik_amd/csrc $d/include
for f in vfclik_amd/csrc/vfik_kernel.hip vfclik_amd/csrc/vfik_kernel.h vfclik_amd/csrc/vfik_abi.cpp vfclik_amd/csrc/Makefile include/vfik.h include/vfik_types.h; do
  git -C $root show $rev:$f > $d/$f
done
make -s -j8 -C $d/vfclik_amd/csrc libvfik_hip.so > $d/build.log 2>&1 || { tail -20 $d/build.log; exit 1; }
cp $d/vfclik_amd/csrc/libvfik_hip.so $out
rm -rf $d
echo built $rev into $out
