#!/bin/bash
# on the GPU box: where the distance-query kernel's time goes ("power-limited") -> gpurun_out/k3ev/
#   k3lab: the inner loop ingredient by ingredient (ns AND cycles per MFMA: the clock is what gives)
#   lone_wave / mfma_chain: what one or two waves per SIMD can issue
#   k3_timestamps: per-layer cycles of the production kernel (instrumented build gpurun_tmp/variants/k3ts.so) for its three widths
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/k3ev
mkdir -p $O
cd $R
{ echo "# tools/mfma_chain.bin"; timeout 120 ./tools/mfma_chain.bin; echo; echo "# tools/lone_wave.bin"; timeout 120 ./tools/lone_wave.bin; } > $O/r03_mfma_issue.txt 2>&1
{ echo "# tools/k3lab.bin 1 1 (epilogues)"; timeout 300 ./tools/k3lab.bin 1 1; echo; echo "# tools/k3lab.bin 3 1 (bias reads, barrier, LDS-DMA stream)"; timeout 300 ./tools/k3lab.bin 3 1; echo; echo "# tools/k3lab.bin 2 1 (one wave per SIMD, one or two column sets)"; timeout 300 ./tools/k3lab.bin 2 1; } > $O/r03_k3lab.txt 2>&1
if [ -f $R/gpurun_tmp/variants/k3ts.so ]; then
  for nv in 80000 600 8; do
    echo "# RA_NV=$nv tools/k3_timestamps.py (instrumented K3: -DRA_TIMESTAMPS)"
    RA_NV=$nv RA_LIB_PATH=$R/gpurun_tmp/variants/k3ts.so timeout 300 python3 tools/k3_timestamps.py 2>/dev/null
    echo
  done > $O/r03_k3_timestamps.txt
fi
cat $O/r03_k3_timestamps.txt | grep -E "launch|clock|ReLU net"
