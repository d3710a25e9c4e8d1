#!/bin/bash
# Step selector like the reference's 03_flash_attention_v2_ring/run.sh:
#   ./run.sh 1            RCCL ring token pass (01_nccl_verify.cu)
#   ./run.sh 2 [args...]  comm/compute overlap and per-link bandwidth probe (02_overlap.cu)
#   ./run.sh 3 [N d]      single-GPU forward against the naive attention on the ring test's data (03_attention_1GPU.cu)
#   ./run.sh 4 [args...]  ring attention test (04_ring_attention.cu)
#   ./run.sh              all of them, in the reference's order (its step 0, an MPI vector add, has no counterpart: no MPI)
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
make -s -C "$HERE/../../../cuda_flashattention_amd/csrc" all ring
make -s -C "$HERE/.." all
step=$1; [ $# -gt 0 ] && shift
if [ -z "$step" ] || [ "$step" -eq 1 ]; then "$HERE/../bin/01_rccl_verify"; fi
if [ -z "$step" ]; then "$HERE/../bin/02_overlap"; elif [ "$step" -eq 2 ]; then "$HERE/../bin/02_overlap" "$@"; fi
if [ -z "$step" ]; then "$HERE/../bin/03_attention_1GPU"; elif [ "$step" -eq 3 ]; then "$HERE/../bin/03_attention_1GPU" "$@"; fi
if [ -z "$step" ]; then "$HERE/../bin/04_ring_attention"; elif [ "$step" -eq 4 ]; then "$HERE/../bin/04_ring_attention" "$@"; fi
