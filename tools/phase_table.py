#!/usr/bin/env python3
"""Per-phase table of a traced run: python tools/phase_table.py <dir or marker csv of rocprofv3 --marker-trace> [bench line json]
The run must have been started with PDLP_ROCTX=2 (ranges synchronise the stream at both ends: a range's wall time is the GPU time
of its work), e.g.
    PDLP_ROCTX=2 rocprofv3 --marker-trace --output-format csv -d out -o m -- python3 bench.py --no-cpu-baseline --solve-tol 0
Prints a markdown table (phase, count, mean ms, total ms, share) and, when the bench line is given, compares the traced cost of a
restart check with the line's timing.check_ms -- the counterpart of the reference's Timer summary (PDLP/util.py:6-27)."""
import csv, glob, json, os, re, sys
from collections import OrderedDict


def find_csv(path):
    if os.path.isfile(path):
        return path
    c = [f for f in glob.glob(os.path.join(path, "**", "*.csv"), recursive=True) if "marker" in os.path.basename(f)]
    if not c:
        raise SystemExit(f"no marker trace csv under {path}")
    return sorted(c, key=os.path.getsize)[-1]


def main():
    rows = list(csv.DictReader(open(find_csv(sys.argv[1]))))
    if not rows:
        raise SystemExit("empty marker trace")
    cols = rows[0].keys()
    name_col = next(c for c in cols if c.lower() in ("function", "name", "message", "marker"))
    s_col = next(c for c in cols if "start" in c.lower())
    e_col = next(c for c in cols if "end" in c.lower())
    phases = OrderedDict()
    for r in rows:
        name = r[name_col]
        if "pdlp:" not in name:
            continue
        name = name[name.index("pdlp:") + 5:].strip().strip('"')
        key = re.sub(r"^Ruiz sweep \d+", "Ruiz sweep", name)
        d = (int(r[e_col]) - int(r[s_col])) / 1e6
        p = phases.setdefault(key, [0, 0.0])
        p[0] += 1
        p[1] += d
    # nested ranges (KKT passes inside a restart check, the exact refresh inside restart work) are listed but not added to the total
    nested = ("KKT pass", "exact products")
    total = sum(v[1] for k, v in phases.items() if not k.startswith(nested))
    print("| phase | ranges | mean ms | total ms | share of the traced time |")
    print("|---|---|---|---|---|")
    for k, (n, t) in phases.items():
        share = "(inside the phases above)" if k.startswith(nested) else f"{100 * t / total:.1f} %"
        print(f"| {k} | {n} | {t / n:.3f} | {t:.1f} | {share} |")
    if len(sys.argv) > 2:
        line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
        chk = phases.get("restart check (3 KKT evaluations)")
        rst = phases.get("restart work (restart, primal weight, termination test)", [0, 0.0])
        if chk:
            per_check = (chk[1] + rst[1]) / chk[0]
            print(f"\ntraced cost of a restart check incl. the restart work at the rate restarts fired: {per_check:.3f} ms "
                  f"({chk[0]} checks, {rst[0]} restarts); the bench line's timing.check_ms: {line['timing']['check_ms']} ms")


if __name__ == "__main__":
    main()
