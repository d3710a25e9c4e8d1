#!/bin/bash
# Builds a variant of libfa2_mi355x.so with extra -D flags into var/<name>.so
# (dev aid for A/B runs: FA2_LIB_PATH=<that file> python tools/gpu_perf.py).
set -e
name=$1; shift
cd "$(dirname "$0")/../cuda_flashattention_amd/csrc"
mkdir -p ../../var /tmp/fa2_var_$name
for f in *.hip fa2_capi.cpp; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I../../include "$@" -x hip -c $f -o /tmp/fa2_var_$name/${f%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../var/$name.so /tmp/fa2_var_$name/*.o
echo built var/$name.so
