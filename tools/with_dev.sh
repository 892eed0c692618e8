#!/bin/bash
# Build the development library on the GPU box (it does not travel: .gpurunignore) and run a command against it.
#   tools/with_dev.sh python tools/ab_forward.py ...
set -e
cd "$(dirname "$0")/.."
make -C fidelityfusion_amd/csrc -j16 dev >/dev/null
export FFGP_LIB="$PWD/fidelityfusion_amd/libffgp_dev.so"
"$@"
