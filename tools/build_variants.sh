#!/bin/bash
# Build A/B variants of libflexlight_hip.so into build/variants/ (travels to the GPU box, not committed).
#   tools/build_variants.sh name1 "-DFLAG=1 ..." name2 "..." ...
set -e
cd "$(dirname "$0")/../web-ray-tracer_amd/csrc"
mkdir -p ../../build/variants
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fPIC -Wno-unused-function \
    -I../../include -I. -I/opt/rocm/include $flags -shared -o ../../build/variants/lib_$name.so flx_api.hip flx_group.hip flx_kernels.hip flx_wavefront.hip flx_walkq.hip flx_walkcoop.hip flx_filter.hip flx_mesh.hip \
    -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib &
done
wait
ls -la ../../build/variants/
