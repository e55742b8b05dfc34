#!/bin/bash
# FETCH_SIZE of the shard product with the panel group as the fast (shipped) or slow (rounds 1-4) index of blockIdx
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in shipped rowblock_fast; do
  lib=torchpdlp_amd/libpdlp_hip.so; [ $v == rowblock_fast ] && lib=tools/_bin/libpdlp_rowblock_fast.so
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/r05/pmc_shard_${v}_$c
    PDLP_LIB=$lib WORLD=8 AG_LIST=0.1 CHUNKS=1 ITERS=10 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r05/pmc_shard_${v}_$c -o p -- python3 tools/shard_iter_timing.py > gpurun_out/r05/pmc_shard_${v}_$c.log 2>&1
  done
  python3 tools/summarize_pmc.py gpurun_out/r05/pmc_shard_${v}_FETCH_SIZE gpurun_out/r05/pmc_shard_${v}_WRITE_SIZE > gpurun_out/r05/pmc_shard_$v.json
  find gpurun_out/r05/pmc_shard_${v}_* -name '*.csv' -size +1M -delete 2>/dev/null
done
python3 - <<'PY'
import json
for v in ("shipped","rowblock_fast"):
    d=json.load(open(f"gpurun_out/r05/pmc_shard_{v}.json"))
    for k,b in d["hbm_bytes_per_launch"].items():
        if "tiled" in k or "rowsum" in k: print(v, k, round(b/1e6,1), "MB", d["FETCH_SIZE"][k]["launches"])
PY
