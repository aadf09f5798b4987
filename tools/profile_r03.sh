#!/bin/bash
# Round-3 profiles (run on the GPU box through gpurun; summaries are then copied into profiles/ by tools/profile_r03_collect.sh).
# One rocprofv3 run per pass; counters in runs of their own (--kernel-trace only beside --pmc).
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
export PYTHONPATH=$PWD TMPDIR=/tmp
O=gpurun_out/prof_r03
mkdir -p $O
B="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-others --min-region-ms 0"
st() { timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o $1 -- python3 "${@:2}" > $O/$1.json 2> $O/$1.err || { tail -3 $O/$1.err; exit 1; }; echo "stats $1 done"; }
pm() { timeout -s KILL 200 rocprofv3 --pmc $2 --kernel-trace -d $O -o $1 -- python3 "${@:3}" > $O/$1.json 2> $O/$1.err || { tail -3 $O/$1.err; exit 1; }; echo "pmc $1 done"; }
st cfg3 $B
st cfg3p $B --output-beam 10
st cfg4den $B --workload cfg4_mmictc_bpe5k_den
st cfg4den3 $B --workload cfg4_3state-skip_bpe5k_den
st cfg2den $B --workload cfg2_2state_char_den
st cfg2num $B --workload cfg2_2state_char_num
st cfg5 tools/search_bench.py 1000000
st fresh tools/time_fresh.py
st search2k tools/search_active_sweep.py 2.6 2.3 2.0
st latw tools/time_lattice_words.py 10 2.6 2.3
pm fetch FETCH_SIZE $B
pm write WRITE_SIZE $B
pm fetchp FETCH_SIZE $B --output-beam 10
pm writep WRITE_SIZE $B --output-beam 10
pm fetch4 FETCH_SIZE $B --workload cfg4_mmictc_bpe5k_den
pm write4 WRITE_SIZE $B --workload cfg4_mmictc_bpe5k_den
pm fetch43 FETCH_SIZE $B --workload cfg4_3state-skip_bpe5k_den
pm write43 WRITE_SIZE $B --workload cfg4_3state-skip_bpe5k_den
pm fetch5 FETCH_SIZE tools/bench_viterbi.py 1000000
pm write5 WRITE_SIZE tools/bench_viterbi.py 1000000
for n in cfg3 cfg3p cfg4den cfg4den3 cfg2den cfg2num cfg5 fresh search2k latw; do python3 tools/prof_summary.py $O/${n}_results.db > $O/${n}_kernel_stats.txt || exit 1; done
python3 tools/pmc_traffic.py $O/fetch_results.db $O/write_results.db "cfg3_ctc_bpe5k_num B=32 T=500" > $O/pmc_traffic.json || exit 1
python3 tools/pmc_traffic.py $O/fetchp_results.db $O/writep_results.db "cfg3_ctc_bpe5k_num B=32 T=500 output_beam=10" > $O/pmc_traffic_cfg3_pruned.json || exit 1
python3 tools/pmc_traffic.py $O/fetch4_results.db $O/write4_results.db "cfg4_mmictc_bpe5k_den B=32 T=500" > $O/pmc_traffic_cfg4den.json || exit 1
python3 tools/pmc_traffic.py $O/fetch43_results.db $O/write43_results.db "cfg4_3state-skip_bpe5k_den B=32 T=500" > $O/pmc_traffic_cfg4den3.json || exit 1
python3 tools/pmc_traffic.py $O/fetch5_results.db $O/write5_results.db "cfg5 Viterbi S=1e6 A=3e6 B=8 T=500 (tools/bench_viterbi.py)" > $O/pmc_traffic_cfg5.json || exit 1
rm -f $O/*.db
find $O -type f -size +2M -delete
ls -la $O | head -80
