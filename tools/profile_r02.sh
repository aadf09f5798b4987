#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun; summaries are then copied into profiles/ by tools/profile_r02_collect.sh).
# One rocprofv3 run per pass; counters in runs of their own (--kernel-trace only beside --pmc).
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
export PYTHONPATH=$PWD TMPDIR=/tmp
O=gpurun_out/prof_r02
mkdir -p $O
B="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-others --min-region-ms 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o cfg3 -- python3 $B > $O/cfg3.json 2> $O/cfg3.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o cfg4den -- python3 $B --workload cfg4_mmictc_bpe5k_den > $O/cfg4den.json 2> $O/cfg4den.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o cfg4den3 -- python3 $B --workload cfg4_3state-skip_bpe5k_den > $O/cfg4den3.json 2> $O/cfg4den3.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o cfg2den -- python3 $B --workload cfg2_2state_char_den > $O/cfg2den.json 2> $O/cfg2den.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o cfg2num -- python3 $B --workload cfg2_2state_char_num > $O/cfg2num.json 2> $O/cfg2num.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o cfg5 -- python3 tools/search_bench.py 1000000 > $O/cfg5.log 2> $O/cfg5.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o fetch -- python3 $B > $O/fetch.json 2> $O/fetch.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O -o write -- python3 $B > $O/write.json 2> $O/write.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o fetch4 -- python3 $B --workload cfg4_mmictc_bpe5k_den > $O/fetch4.json 2> $O/fetch4.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O -o write4 -- python3 $B --workload cfg4_mmictc_bpe5k_den > $O/write4.json 2> $O/write4.err || exit 1
for n in cfg3 cfg4den cfg4den3 cfg2den cfg2num cfg5; do python3 tools/prof_summary.py $O/${n}_results.db > $O/${n}_kernel_stats.txt || exit 1; done
python3 tools/pmc_traffic.py $O/fetch_results.db $O/write_results.db "cfg3_ctc_bpe5k_num B=32 T=500" > $O/pmc_traffic.json || exit 1
python3 tools/pmc_traffic.py $O/fetch4_results.db $O/write4_results.db "cfg4_mmictc_bpe5k_den B=32 T=500" > $O/pmc_traffic_cfg4den.json || exit 1
rm -f $O/*.db
find $O -type f -size +2M -delete
ls -la $O | head -60
