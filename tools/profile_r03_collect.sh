#!/bin/bash
# Copies the summaries tools/profile_r03.sh left under gpurun_out/prof_r03 into profiles/ (tracked) and stamps them with the
# hash of the kernel source they were measured on (bench.py nulls the fields it reads from them when the source moves on).
cd "$(dirname "$0")/.." || exit 1
S=gpurun_out/prof_r03
cp $S/cfg3_kernel_stats.txt profiles/r03_bench_cfg3_kernel_stats.txt
cp $S/cfg3p_kernel_stats.txt profiles/r03_bench_cfg3_pruned_beam10_kernel_stats.txt
cp $S/cfg4den_kernel_stats.txt profiles/r03_bench_cfg4_mmictc_den_kernel_stats.txt
cp $S/cfg4den3_kernel_stats.txt profiles/r03_bench_cfg4_3state-skip_den_kernel_stats.txt
cp $S/cfg2den_kernel_stats.txt profiles/r03_bench_cfg2_den_kernel_stats.txt
cp $S/cfg2num_kernel_stats.txt profiles/r03_bench_cfg2_num_kernel_stats.txt
cp $S/cfg5_kernel_stats.txt profiles/r03_search_cfg5_S1e6_kernel_stats.txt
cp $S/cfg5.json profiles/r03_search_cfg5_S1e6_timings.txt
cp $S/fresh_kernel_stats.txt profiles/r03_fresh_graphs_kernel_stats.txt
cp $S/fresh.json profiles/r03_fresh_graphs_timings.txt
cp $S/search2k_kernel_stats.txt profiles/r03_search_active_sweep_kernel_stats.txt
cp $S/search2k.json profiles/r03_search_active_sweep_timings.txt
cp $S/latw_kernel_stats.txt profiles/r03_lattice_words_kernel_stats.txt
cp $S/latw.json profiles/r03_lattice_words_timings.txt
cp $S/pmc_traffic.json profiles/r03_pmc_traffic.json
cp $S/pmc_traffic_cfg3_pruned.json profiles/r03_pmc_traffic_cfg3_pruned_beam10.json
cp $S/pmc_traffic_cfg4den.json profiles/r03_pmc_traffic_cfg4_mmictc_den.json
cp $S/pmc_traffic_cfg4den3.json profiles/r03_pmc_traffic_cfg4_3state-skip_den.json
cp $S/pmc_traffic_cfg5.json profiles/r03_pmc_traffic_cfg5_viterbi_S1e6.json
python3 - <<'PY'
import hashlib, json
h = hashlib.sha256()
for f in ("waterfall_amd/csrc/wf_kernels.hip", "waterfall_amd/csrc/wf_internal.h"):
    h.update(open(f, "rb").read())
json.dump({"kernel_source_sha256": h.hexdigest(),
           "commands": "tools/profile_r03.sh (rocprofv3 --kernel-trace --stats per workload; --pmc FETCH_SIZE / WRITE_SIZE in runs of their own)",
           "bench_args": "--steps 20 --warmup 5 --no-cpu-baseline --no-others --min-region-ms 0"},
          open("profiles/r03_meta.json", "w"), indent=1)
print(open("profiles/r03_meta.json").read())
PY
