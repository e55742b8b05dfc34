#!/usr/bin/env python3
"""mean value per launch of every counter rocprofv3 collected, per library kernel: summarize_counters.py <dir with pass dirs>"""
import csv, glob, json, os, re, sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    m = re.search(r"(k_\w+)<(float|double)(?:, (?:float|double))?, \(anonymous namespace\)::(\w+)<[^>]*>", name)
    if m:
        return f"{m.group(1)}<{m.group(2)},{m.group(3)}>"
    m2 = re.search(r"(k_\w+)", name)
    return m2.group(1) if m2 else None


acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k is None:
            continue
        a = acc[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
out = {k: {c: round(s / n, 1) for c, (n, s) in sorted(v.items())} | {"launches": max(n for n, _ in v.values())} for k, v in acc.items()}
keep = {k: v for k, v in out.items() if k.startswith(("k_tiled", "k_csr", "k_rowsum"))}
print(json.dumps(keep, indent=1))
