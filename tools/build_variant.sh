#!/bin/bash
# Build a VARIANT of libvfik_hip.so out of tree (never in tree: the in-tree source-hash stamp must describe the product):
#   tools/build_variant.sh <name> [extra compiler flags, e.g. -DVFIK_NT_MIN_NJ=0] [-- patch-file ...]
#   (make variables through the environment: VFIK_MAKE_ARGS='SCHED=-mllvm\ -amdgpu-sched-strategy=iterative-ilp')
# The library lands in tools/variants/<name>.so (git-ignored, travels to the GPU box) for tools/ab_compare.py.
set -eu
name=$1; shift
flags=()
patches=()
while [ $# -gt 0 ]; do
  if [ "$1" = "--" ]; then shift; patches=("$@"); break; fi
  flags+=("$1"); shift
done
R=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/vfik_variant_$name
rm -rf "$W"; mkdir -p "$W/vfclik_amd" "$R/tools/variants"
cp -r "$R/include" "$W/include"
cp -r "$R/vfclik_amd/csrc" "$W/vfclik_amd/csrc"
rm -f "$W"/vfclik_amd/csrc/*.o "$W"/vfclik_amd/csrc/*.so "$W"/vfclik_amd/csrc/*.srchash
for p in "${patches[@]:-}"; do [ -n "$p" ] && (cd "$W" && patch -p1 < "$p"); done
eval make -s -j6 -C "$W/vfclik_amd/csrc" libvfik_hip.so "CXXFLAGS='-O3 -std=c++17 -fPIC -Wall -Wno-unused-result ${flags[*]:-}'" ${VFIK_MAKE_ARGS:-}
cp "$W/vfclik_amd/csrc/libvfik_hip.so" "$R/tools/variants/$name.so"
echo "built tools/variants/$name.so with: ${flags[*]:-} ${patches[*]:-}"
