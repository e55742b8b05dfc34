#!/bin/bash
# Collects what profiles/ holds for one bench configuration, on the GPU box:
#   tools/profile_bench.sh <tag> [bench.py args...]     e.g.  tools/profile_bench.sh 10Mx10M_tiled
# 1. rocprofv3 --kernel-trace --stats          -> gpurun_out/prof_<tag>/stats_kernel_stats.csv
# 2. rocprofv3 --kernel-trace --pmc FETCH_SIZE -> counters of every launch   (separate passes, as the
# 3. rocprofv3 --kernel-trace --pmc WRITE_SIZE                                MI355X guide prescribes)
# 4. tools/summarize_pmc.py                    -> gpurun_out/prof_<tag>/pmc.json
# The program follows "--" directly (no env/bash hop: the profiler has initialised the GPU by then).
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/prof_$tag
rm -rf "$out" && mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o stats -- python3 bench.py --no-cpu-baseline --steps 80 --warmup 40 "$@" > "$out/stats.log" 2>&1
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o fetch -- python3 bench.py --no-cpu-baseline --steps 40 --warmup 3 "$@" > "$out/fetch.log" 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -o write -- python3 bench.py --no-cpu-baseline --steps 40 --warmup 3 "$@" > "$out/write.log" 2>&1
echo "write done"
python3 tools/summarize_pmc.py "$out/fetch" "$out/write" > "$out/pmc.json"
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"
find "$out" -name '*.csv' ! -name 'kernel_stats.csv' -size +1M -delete      # the raw traces are large; the summaries are what is kept
tail -1 "$out/stats.log"
cat "$out/pmc.json"
