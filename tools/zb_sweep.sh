for zb in 16 22 28 34; do
  WF_ZERO_BLOCKS=$zb timeout -k 10 100 python bench.py --no-others --no-cpu-baseline --steps 200 > gpurun_out/zb_$zb.json 2>/dev/null || exit 1
done
