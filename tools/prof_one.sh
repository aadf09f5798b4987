#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py run on the GPU box; usage: tools/prof_one.sh <tag> <bench args...>
# leaves gpurun_out/prof/<tag>_kernel_stats.txt and <tag>.json
set -o pipefail
cd "$(dirname "$0")/.." || exit 1
export PYTHONPATH=$PWD TMPDIR=/tmp
O=gpurun_out/prof
mkdir -p $O
tag=$1; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o $tag -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-others --min-region-ms 0 "$@" > $O/$tag.json 2> $O/$tag.err || { tail -5 $O/$tag.err; exit 1; }
python3 tools/prof_summary.py $O/${tag}_results.db > $O/${tag}_kernel_stats.txt || exit 1
rm -f $O/${tag}_results.db
head -12 $O/${tag}_kernel_stats.txt | cut -c1-150
