#!/usr/bin/env python3
"""Markdown tables of one bench campaign (bench_detail.json): the headline, every same-run variant and the other decks with the per-iteration
and the per-time-step figures side by side.   python tools/results_table.py profiles/r04_r_bench_detail.json"""
import json, sys
d = json.load(open(sys.argv[1]))


def row(name, v):
    if "failed" in v:
        return "| `%s` | failed: %s |" % (name, v["failed"][:60])
    p = v.get("per_time_step") or {}
    f = lambda x, n=2: "-" if x is None else ("%." + str(n) + "f") % x
    return "| `%s` | %s | %s | %s | %s | %s | %s | %s | %s | %s |" % (
        name, f(v["value"], 1), f(v.get("value_mean_solving"), 1), f(v.get("ms_per_solving_iteration_median"), 3), f(v.get("linear_iterations_per_solving_iteration"), 2),
        f(p.get("newton_iterations_per_time_step"), 2), f(p.get("ms_per_converged_time_step"), 2), f(p.get("ms_per_simulated_day"), 3),
        p.get("chopped_attempts", "-"), p.get("time_steps", "-"))


head = ("| configuration | Mcell-upd/s (median) | (mean over solving calls) | ms per solving iteration | linear its per solve | Newton its per time step | "
        "ms per converged time step | ms per simulated day | chopped attempts | time steps in the window |\n|---|---|---|---|---|---|---|---|---|---|")
print(head)
print(row("HEADLINE " + d["line"]["config"]["linear_solver"], d["main"]))
for k, v in d["same_run_variants"].items():
    print(row(k, v))
print()
print(head)
for k, v in d["other_decks"].items():
    print(row(k, v))
