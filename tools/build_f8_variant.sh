#!/bin/bash
# Dev aid: builds var/f8_<name>.so = the product library with the fp8 forward's bodies regenerated under the environment
# given as VAR=value arguments (tools/gen_fwd_fp8_body.py switches) and extra -D flags after "--".
#   tools/build_f8_variant.sh b64 FA2_GEN_F8_BUDGET=64 -- -DFOO
set -e
name=$1; shift
envs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
[ "$1" == "--" ] && shift
root="$(cd "$(dirname "$0")/.." && pwd)"
tmp=/tmp/fa2_f8_$name; rm -rf $tmp; mkdir -p $tmp "$root/var"
cp "$root"/cuda_flashattention_amd/csrc/*.h "$root"/cuda_flashattention_amd/csrc/*.inc "$root"/cuda_flashattention_amd/csrc/fa2_fwd_fp8.hip $tmp/
(cd "$root/tools" && env "${envs[@]}" python3 gen_fwd_fp8_body.py --out $tmp/fa2_fwd_fp8_body.inc >/dev/null)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -Wno-inline-asm -I"$root/include" "$@" -c $tmp/fa2_fwd_fp8.hip -o $tmp/fa2_fwd_fp8.o
obj="$root/cuda_flashattention_amd/csrc/_obj"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o "$root/var/f8_$name.so" $(ls $obj/*.o | grep -v fa2_fwd_fp8.o | grep -v hooks) $tmp/fa2_fwd_fp8.o
echo built var/f8_$name.so
