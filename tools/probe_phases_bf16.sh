#!/bin/bash
# phase ablation of the one-tile bf16-storage GEMM kernels (FS2_GEMM_PROBE: 1 no MFMA, 2 no DMA, 4 no epilogue)
cd $GRAFT_REPO_ROOT
# needs a probe build: FS2_BUILD_PROBES=1 python -m fastspeech2_lightning_amd.build --force (rebuild without it afterwards)
for t in 20 22; do
  for pr in 0 1 2 4 3 5 6 7; do
    echo "tile $t probe $pr: $(FS2_GEMM_PROBE=$pr timeout -k 10 100 python tools/bench_epilogue_bf16.py $t 2>/dev/null | grep -E 'store bf16|silu \+ dropout \+ pre-activation bf16|residual' | awk '{printf "%s ", $0}')"
  done
done
