#!/bin/bash
# Hardware counters of the round-4 LDS-resident prototypes (tools/src/lds_proto*.hip): one rocprofv3 pass per counter set.
#   tools/proto_counters.sh   ->  gpurun_out/proto_pmc/<binary>_<pass>/..., summary printed
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/proto_pmc
rm -rf "$out" && mkdir -p "$out"
for bin in lds_proto2_d4 lds_proto3 lds_proto5; do
  i=0
  for s in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_TA_BUSY_sum"; do
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $s --output-format csv -d "$out/${bin}_p$i" -o p -- tools/_bin/$bin > "$out/${bin}_p$i.log" 2>&1 || { echo "$bin pass $i failed"; tail -3 "$out/${bin}_p$i.log"; }
    i=$((i+1))
  done
done
python3 - <<'PY'
import csv, glob, os, re, json
from collections import defaultdict
csv.field_size_limit(1 << 30)
res = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob("gpurun_out/proto_pmc/**/*counter_collection.csv", recursive=True):
    tag = f.split("/")[2].rsplit("_p", 1)[0]
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)<([^>]*)>", r["Kernel_Name"])
        if not m: continue
        a = res[f"{tag}:{m.group(1)}<{m.group(2)}>"][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: round(s / n, 1) for c, (n, s) in sorted(v.items())} for k, v in sorted(res.items())}
json.dump(out, open("gpurun_out/proto_pmc/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
find "$out" -name '*.csv' -size +1M -delete
