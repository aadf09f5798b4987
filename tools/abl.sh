#!/bin/bash
# Ablation timing of the log-semiring sweep_kernel on cfg4's denominators (WF_SWEEP_ABLATE bits: 32 no row stores,
# 64 no heavy states, 128 no light states; timing only, results are wrong).  Run on the GPU box from the repo root.
for a in 0 32 64 128 96 160 192 224; do
  echo "ablate $a"
  WF_NO_LINBIG_SWEEP=1 WF_SWEEP_ABLATE=$a timeout -k 10 100 python bench.py --workload cfg4_mmictc_bpe5k_den --steps 5 --warmup 2 --no-cpu-baseline --no-others > gpurun_out/abl.json \
    && python tools/show_bench.py gpurun_out/abl.json | grep sweep | head -1 || exit 1
done
