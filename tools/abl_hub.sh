# timing ablations of sweep_hub_kernel on cfg4's mmictc denominator (WF_SWEEP_ABLATE: 1 = no row stores, 2 = no loads)
for a in 0 1 2 3; do
  WF_SWEEP_ABLATE=$a timeout -k 10 100 python bench.py --workload cfg4_mmictc_bpe5k_den --steps 5 --warmup 2 --no-cpu-baseline --no-others > gpurun_out/ablh_$a.json 2>/dev/null || exit 1
done
