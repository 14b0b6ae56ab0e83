#!/bin/bash
# builds everything that travels, then runs a script on the GPU box: tools/gpu.sh SCRIPT [TIMEOUT_S]  (log: /tmp/gpu_<script>.log)
set -e
cd "$(dirname "$0")/.."
make -s -j8 product emu
make -s -C oracle all
S=$1; T=${2:-1800}
/usr/local/graft/bin/gpurun --timeout $T -- "bash tools/$S" > /tmp/gpu_${S%.sh}.log 2>&1
