#!/bin/bash
# usage: bash scripts/build_variant.sh <tag> [extra hipcc -D flags...]  ->  opencl_render_amd/variants/lib_<tag>.so
# Same sources and exactness flags as csrc/Makefile; only rt_wavefront.hip is recompiled with the extra defines.
set -e
tag=$1; shift
cd "$(dirname "$0")/../opencl_render_amd/csrc"
make -s
mkdir -p build ../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-gpu-rdc \
    -I../../include -I. "$@" -c rt_wavefront.hip -o build/wf_$tag.o
/opt/rocm/bin/hipcc -shared -fPIC -o ../variants/lib_$tag.so build/rt_kernels.o build/wf_$tag.o build/rt_build_device.o build/rt_kat.o build/rt_scene_prep.o build/rt_api.o build/rt_builders.o build/rt_mathabi.o build/rt_frontend.o build/rt_fileio.o -pthread
echo built opencl_render_amd/variants/lib_$tag.so
