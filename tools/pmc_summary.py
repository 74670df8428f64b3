#!/usr/bin/env python3
"""Per-kernel averages of the PMC passes written by tools/profile_round.sh (rocprofv3 counter_collection CSVs) and of the kernel trace."""
import csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(out, "pmc*_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if "k_assemble_rows" in k or "k_spmv" in k:          # keep the template arguments that tell float from double apart
                k = row["Kernel_Name"].split("(")[0]
            amg_pass = os.path.basename(d).startswith("pmcamg_")
            wanted = ("k_amg_galerkin", "k_amg_row_sub", "k_cpr_rows", "k_dense_invert", "k_ilu_factor", "k_amg_residual", "k_cpr_sum_eqs", "k_cpr_presidual") if amg_pass else \
                     ("k_assemble_rows", "k_cell_values", "k_flux", "k_cell_props", "k_spmv", "k_conv_partial")
            if not any(t in k for t in wanted):
                continue
            if amg_pass:
                k = row["Kernel_Name"].split("(")[0]
            key = (k, row["Counter_Name"])
            a = acc.setdefault(key, [0.0, set()])
            a[0] += float(row["Counter_Value"]); a[1].add(row["Dispatch_Id"])
        for (k, c), (v, ids) in acc.items():
            res.setdefault(k, {})[c] = v / max(1, len(ids))
            res[k]["launches"] = len(ids)
for k, r in res.items():
    if "FETCH_SIZE" in r:
        r["fetched_MB_x2"] = round(2 * r["FETCH_SIZE"] * 1024 / 1e6, 1)       # KB reported; gfx950 tallies 64 B per 128-B request (MI355X_MICROARCH.md)
    if "WRITE_SIZE" in r:
        r["written_MB"] = round(r["WRITE_SIZE"] * 1024 / 1e6, 1)
    if "TCC_HIT_sum" in r and "TCC_MISS_sum" in r:
        r["l2_hit_rate"] = round(r["TCC_HIT_sum"] / (r["TCC_HIT_sum"] + r["TCC_MISS_sum"]), 3)
print(json.dumps(res, indent=1, sort_keys=True))
