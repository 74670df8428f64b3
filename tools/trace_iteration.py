#!/usr/bin/env python3
"""One Newton iteration out of a rocprofv3 --kernel-trace CSV: start offset, duration, queue, short kernel name.
    python tools/trace_iteration.py <run_kernel_trace.csv> [which=-3] [marker=k_cell_values]     (which: index of the iteration, negative from the end)"""
import csv
import re
import sys

path = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
marker = sys.argv[3] if len(sys.argv) > 3 else "k_cell_values"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if marker in r[3]]
if len(sys.argv) > 4:          # the LAST iteration that contains a kernel whose name has argv[4] (e.g. k_ilu_lower: an iteration with a linear solve)
    need = sys.argv[4]
    bounds = starts + [len(rows)]
    cand = [k for k in range(len(starts) - 1) if any(need in r[3] for r in rows[bounds[k]:bounds[k + 1]])]
    which = cand[which if which < 0 else min(which, len(cand) - 1)]
a = starts[which]
b = starts[which + 1] if which + 1 < len(starts) and which + 1 != 0 else len(rows)
qs = {}


def short(n):
    n = re.sub(r"^void\s+", "", n)
    n = re.sub(r"opmgpu::", "", n)
    return re.sub(r"\(.*$", "", n)


t0 = rows[a][0]
print("one Newton iteration (%s .. next %s): span %.1f us under the profiler" % (marker, marker, (rows[b - 1][1] - t0) / 1e3))
tot = {}
for s, e, q, n in rows[a:b]:
    qs.setdefault(q, "q%d" % (len(qs) + 1))
    nm = short(n)
    tot[nm] = tot.get(nm, 0.0) + (e - s) / 1e3
    print("%9.1f us %8.1f us  %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, qs[q], nm))
print("-- totals by kernel (us)")
for nm, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("%9.1f  %s" % (v, nm))
