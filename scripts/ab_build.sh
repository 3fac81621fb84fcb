#!/bin/bash
# scripts/ab_build.sh NAME [extra hipcc flags]: builds the current kernel sources into ab/libqdsim_NAME.so (A/B runs: QDSIM_LIB=ab/libqdsim_NAME.so)
set -e
cd "$(dirname "$0")/.."
mkdir -p ab
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -fno-fast-math -Iinclude -Irl-agent-for-qubit-array-tuning_amd/csrc "$@" \
    -o ab/libqdsim_$name.so rl-agent-for-qubit-array-tuning_amd/csrc/qd_api.hip
echo built ab/libqdsim_$name.so
