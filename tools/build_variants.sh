#!/bin/bash
# Build A/B variants of libflexlight_hip.so into build/variants/ (travels to the GPU box, not committed), with the Makefile's own
# per-file flags plus the variant's:   tools/build_variants.sh name1 "-DFLAG=1 ..." name2 "..." ...
set -e
cd "$(dirname "$0")/../web-ray-tracer_amd/csrc"
mkdir -p ../../build/variants
names=()
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  names+=("$name")
  make -s -j4 OBJDIR=../../build/vobj_$name OUT=../../build/variants/lib_$name.so EXTRA="$flags" > ../../build/variants/$name.build.log 2>&1 &
done
wait
for name in "${names[@]}"; do rm -rf "../../build/vobj_$name"; done
ls -la ../../build/variants/*.so
