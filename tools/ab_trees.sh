#!/bin/bash
# Same-box A/B of the working tree against another git revision (the method that caught round 4's register-spill regression: boxes of
# the pool differ by +-3 %, so two bench lines from two gpurun calls cannot tell a 5 % change from noise).
#   HERE (build container):  tools/ab_trees.sh prepare <git-ref>      builds <git-ref> into tools/_bin/_reftree (travels with gpurun)
#   THERE (GPU box):         gpurun -- 'bash tools/ab_trees.sh run [bench.py args...] > gpurun_out/ab_trees.log 2>&1'
#   afterwards:              rm -rf tools/_bin/_reftree
set -e
cd "$(dirname "$0")/.."
if [ "$1" == "prepare" ]; then
  rm -rf tools/_bin/_reftree && mkdir -p tools/_bin/_reftree
  git archive "$2" | tar -x -C tools/_bin/_reftree
  (cd tools/_bin/_reftree && bash torchpdlp_amd/csrc/build.sh > /dev/null && make -s -C oracle)
  echo "reference tree $2 built in tools/_bin/_reftree"
  exit 0
fi
shift || true
for rep in 1 2; do
  for tree in . tools/_bin/_reftree; do
    (cd $tree && python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --solve-tol 0 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tree', d['value'], 'it/s; kernels', d['roofline'].get('launch_ms'), d['roofline'].get('other_kernel_ms'), 'ms; check', d['timing']['check_ms'], 'ms')")
  done
done
