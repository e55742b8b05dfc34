#!/bin/bash
# Round-5 measurement batch for profiles/ (run on the GPU box through gpurun; every step appends to gpurun_out/r05_measure.log).
#   tools/r05_measure.sh [10m] [1m] [neos3] [shard] [phases]      (default: all)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
what="${*:-10m 1m neos3 shard phases}"
log=gpurun_out/r05_measure.log
say() { echo "[$(date +%T)] $*" | tee -a "$log"; }
mkdir -p gpurun_out
if [[ "$what" == *10m* ]]; then
  say "bench 10M default"
  timeout -k 10 400 python3 bench.py > gpurun_out/r05_bench_10Mx10M.json 2> gpurun_out/r05_bench_10Mx10M.err || say "bench 10M failed"
  say "bench 10M, the driver's flags (--steps 20 --warmup 5)"
  timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_10Mx10M_steps20.json 2> gpurun_out/r05_bench_10Mx10M_steps20.err || say "bench 10M steps20 failed"
  say "profile 10M (stats + FETCH + WRITE)"
  timeout -k 10 700 bash tools/profile_bench.sh r05_10Mx10M_tiled >> "$log" 2>&1 || say "profile 10M failed"
fi
if [[ "$what" == *1m* ]]; then
  export PDLP_BENCH_N=1000000 PDLP_BENCH_NNZ=5
  say "bench 1M x 1M, 5 per row"
  timeout -k 10 200 python3 bench.py --steps 2000 --warmup 200 > gpurun_out/r05_bench_1Mx1M.json 2> gpurun_out/r05_bench_1Mx1M.err || say "bench 1M failed"
  say "profile 1M (stats + FETCH + WRITE)"
  timeout -k 10 300 bash tools/profile_bench.sh r05_1Mx1M --solve-tol 0 >> "$log" 2>&1 || say "profile 1M failed"
  say "L2 counters 1M"
  timeout -k 10 300 bash tools/pmc_sets.sh r05_1M_l2 "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" -- --solve-tol 0 >> "$log" 2>&1 || say "L2 counters failed"
  unset PDLP_BENCH_N PDLP_BENCH_NNZ
fi
if [[ "$what" == *neos3* ]]; then
  say "neos3-shaped"
  timeout -k 10 200 python3 tools/bench_neos3_shape.py > gpurun_out/r05_neos3_shape.log 2>&1 || say "neos3 failed"
  cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/prof_r05_neos3_$c
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/prof_r05_neos3_$c -o p -- python3 tools/bench_neos3_shape.py > gpurun_out/prof_r05_neos3_$c.log 2>&1 || say "neos3 $c failed"
  done
  python3 tools/summarize_pmc.py gpurun_out/prof_r05_neos3_FETCH_SIZE gpurun_out/prof_r05_neos3_WRITE_SIZE > gpurun_out/r05_pmc_neos3_shape.json 2>> "$log" || say "neos3 summary failed"
  find gpurun_out/prof_r05_neos3_* -name '*.csv' -size +1M -delete 2>/dev/null
fi
if [[ "$what" == *shard* ]]; then
  say "one rank's shard, whole iterations with stand-in collectives (tools/shard_iter_timing.py)"
  # (PEER=1: the direct exchange in loopback first, then the loop with the stand-in collectives)
  (PEER=1 WORLD=8 AG_LIST=0.1,0.2 CHUNKS=1,2 timeout -k 10 300 python3 tools/shard_iter_timing.py
   PEER=1 WORLD=4 AG_LIST=0.16,0.3 CHUNKS=1,2 timeout -k 10 300 python3 tools/shard_iter_timing.py
   PEER=1 WORLD=2 AG_LIST=0.31 CHUNKS=1 timeout -k 10 300 python3 tools/shard_iter_timing.py
   PEER=1 WORLD=8 AG_LIST=0.1,0.2 CHUNKS=1,2 DTYPE=mixed timeout -k 10 300 python3 tools/shard_iter_timing.py) 2>&1 | grep -v amdgpu > gpurun_out/r05_shard_iter.log || say "shard failed"
  cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
  rm -rf gpurun_out/prof_r05_shard8
  WORLD=8 AG_LIST=0.1 CHUNKS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_shard8 -o s -- python3 tools/shard_iter_timing.py > gpurun_out/r05_shard8_prof.log 2>&1 || say "shard profile failed"
  cp "$(find gpurun_out/prof_r05_shard8 -name '*kernel_stats.csv' | head -1)" gpurun_out/r05_kernel_stats_shard8.csv 2>/dev/null
  find gpurun_out/prof_r05_shard8 -name '*.csv' -size +1M -delete 2>/dev/null
fi
if [[ "$what" == *phases* ]]; then
  say "roctx ranges: per-phase table of a bench run (PDLP_ROCTX=2)"
  cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
  rm -rf gpurun_out/r05_markers
  PDLP_ROCTX=2 timeout -k 10 300 rocprofv3 --marker-trace --output-format csv -d gpurun_out/r05_markers -o m -- python3 bench.py --no-cpu-baseline --solve-tol 0 > gpurun_out/r05_markers_bench.json 2> gpurun_out/r05_markers.err || say "marker trace failed"
  python3 tools/phase_table.py gpurun_out/r05_markers gpurun_out/r05_markers_bench.json > gpurun_out/r05_phase_table_10Mx10M.md 2>> "$log" || say "phase table failed"
fi
say done
