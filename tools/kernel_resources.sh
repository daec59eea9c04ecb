#!/bin/bash
# VGPR / spill / occupancy of every kernel in one translation unit:
#   tools/kernel_resources.sh nmx_wave.hip -DNMX_NT_FIXED=64
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" -I$ROOT/include -I$ROOT/py_neuromodulation_amd/csrc \
  --cuda-device-only -c $ROOT/py_neuromodulation_amd/csrc/$src -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|  VGPRs:|VGPRs Spill|Occupancy|SGPRs Spill" |
  sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - 
