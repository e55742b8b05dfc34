#!/bin/bash
# Hardware counters of the library's kernels for one bench configuration, one rocprofv3 pass per counter set (counters
# only: no trace domains next to --pmc).  tools/pmc_sets.sh <tag> "<set 1>" "<set 2>" ... [-- bench.py args]
# Result: gpurun_out/pmc_<tag>/summary.json = mean counter value per launch of every library kernel.
set -e
tag=$1; shift
sets=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("$1"); shift; done
[ "$1" == "--" ] && shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/pmc_$tag
rm -rf "$out" && mkdir -p "$out"
i=0
for s in "${sets[@]}"; do
  rocprofv3 --kernel-trace --pmc $s --output-format csv -d "$out/p$i" -o p -- python3 bench.py --no-cpu-baseline --steps 40 --warmup 3 "$@" > "$out/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/p$i.log"; }
  echo "pass $i done: $s"
  i=$((i+1))
done
python3 tools/summarize_counters.py "$out" > "$out/summary.json"
find "$out" -name '*.csv' -size +1M -delete
cat "$out/summary.json"
