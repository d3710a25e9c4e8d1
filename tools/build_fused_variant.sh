#!/bin/bash
# Dev aid: builds var/fu_<name>.so = the product library with the single-kernel backward's bodies regenerated under the
# environment given as VAR=value arguments (tools/gen_fused_body.py switches) and extra -D flags after "--".
#   tools/build_fused_variant.sh ailv FA2_GEN_AILV=1 -- '-DFA2_FUSED_DSKEY(row)=(((row)>>1)&7)'
# Linked -Bsymbolic so that several variants can live in ONE process (tools/gpu_ab_multi.py).
set -e
name=$1; shift
envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" == "--" ] && shift
root="$(cd "$(dirname "$0")/.." && pwd)"
tmp=/tmp/fa2_fu_$name; rm -rf $tmp; mkdir -p $tmp "$root/var"
cp "$root"/cuda_flashattention_amd/csrc/*.h "$root"/cuda_flashattention_amd/csrc/*.inc "$root"/cuda_flashattention_amd/csrc/fa2_bwd_fused.hip $tmp/
(cd "$root/tools" && env "${envs[@]}" python3 gen_fused_body.py --out $tmp/fa2_bwd_fused_body.inc >/dev/null)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -Wno-inline-asm -I"$root/include" "$@" -c $tmp/fa2_bwd_fused.hip -o $tmp/fa2_bwd_fused.o
obj="$root/cuda_flashattention_amd/csrc/_obj"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o "$root/var/fu_$name.so" $(ls $obj/*.o | grep -v fa2_bwd_fused.o | grep -v hooks) $tmp/fa2_bwd_fused.o
echo built var/fu_$name.so
