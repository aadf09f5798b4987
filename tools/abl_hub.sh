# timing ablations of sweep_hub_kernel on cfg4's denominators (WF_SWEEP_ABLATE bits: 1 = row stores all to row 0,
# 2 = no loads, 4 = no per-frame barrier, 32 = 4-byte LDS-DMA pieces only, 64 = the general frame body); read
# kernels.sweep.avg_ms (the results of an ablated run are wrong and the fallback redoes them: ms_per_step means nothing)
for w in cfg4_mmictc_bpe5k_den cfg4_3state-skip_bpe5k_den; do
  for a in ${ABL:-0 1 2 3 7 64}; do
    WF_SWEEP_ABLATE=$a timeout -k 10 100 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-others > gpurun_out/ablh_${w}_$a.json 2>/dev/null || exit 1
  done
done
