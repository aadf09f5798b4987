# fresh-graph legs of bench.py at several host-thread counts (the numerator compiler's worker pool)
for nt in 1 2 4 8 16; do
  WF_HOST_THREADS=$nt timeout -k 10 100 python bench.py --no-others --no-cpu-baseline > gpurun_out/ht_$nt.json 2>gpurun_out/ht_$nt.err || exit 1
done
