#!/bin/bash
# Runs tools/exp/mfma_power and samples board power / shader clock once a second next to it (see mfma_power.hip).
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3
( while true; do echo "$(date +%s.%N) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk' | sed 's/.*: //' | tr '\n' ' ')"; sleep 0.7; done ) > gpurun_out/r3/mfma_power_smi.txt &
SMI=$!
tools/exp/mfma_power ${1:-8} $2 | tee gpurun_out/r3/mfma_power.txt
kill $SMI
