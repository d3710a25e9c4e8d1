#!/bin/bash
# Builds (hipcc --offload-arch=gfx950) and runs this directory's test main, like the
# reference's run_local.sh.  No arguments: the reference's own test cases; or: B H N d [causal [iters]]
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
make -s -C "$HERE/../../../cuda_flashattention_amd/csrc" all ring
make -s -C "$HERE/.." all
exec "$HERE/../bin/02_flash_attention_v2_backward" "$@"
