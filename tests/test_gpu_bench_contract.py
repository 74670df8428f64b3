"""GPU: `python bench.py` on a small deck prints ONE JSON line that carries the driver's contract (metric / value / unit / n_gpus / steps /
warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus the `roofline` and `cpu_baseline`
objects, and the same-run variants; the distributed code path with one rank prints the same contract."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--nx", "24", "--ny", "24", "--nz", "12", "--steps", "4", "--warmup", "1", "--rate", "10", *args],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract(gpu_lib):
    d = _run()
    assert d["metric"].startswith("Mcell-updates/sec") and d["unit"] == "Mcell-updates/s"
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "f64" in d["dtype"]
    # value = cells / MEDIAN duration of the timed iterations that include a solve; the mean over all calls stays beside it
    assert d["value"] > 0 and abs(d["value"] - d["config"]["cells"] / (d["ms_per_solving_iteration_median"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    assert abs(d["value_all_calls_mean"] - d["config"]["cells"] / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value_all_calls_mean"]
    calls = d["timed_calls"]
    assert len(calls["ms"]) == 4 and sum(calls["solved"]) == d["config"]["solving_iterations"] >= 1
    import statistics
    assert abs(statistics.median(m for m, s in zip(calls["ms"], calls["solved"]) if s) - d["ms_per_solving_iteration_median"]) < 1e-3
    assert sum(calls["ms"]) <= 1.05 * d["ms_per_step"] * 4 + 0.5           # the per-call durations add up to the timed region
    b = d["breakdown_ms_per_step"]
    assert b["assemble"] + b["linear_solve"] + b["update"] <= d["ms_per_solving_iteration_median"] * 1.02
    # the headline is the reference-runnable configuration: CPR in double (NewtonIterationBlackoilCPR.cpp:117-140) with newton_use_gmres
    assert d["dtype"] == "f64" and d["config"]["linear_solver"] == "cpr(amg V-cycle + ilu0) + gmres(40)" and d["config"]["gmres_true_residual_check"] is False
    assert d["config"]["workload"].startswith("cart24x24x12") and d["config"]["workload"].endswith("_fivespot") and "model" not in d["config"]
    assert d["config"]["time_steps_not_converged"] == 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["traffic"] is None
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["kernel"].startswith("k_spmv")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    v = d["same_run_variants"]
    assert {"reference_default_solver_ilu0_with_wells", "without_wells", "cpr_f64_bicgstab_with_wells", "cpr_f64_gmres_verified_with_wells", "cpr_f64_gmres_fixed_correction_factor_with_wells", "cpr_f32_gmres_with_wells", "dt30_f64_ilu0_with_wells",
            "dt30_f64_cpr_gmres_with_wells", "dt1_f32_ilu0_with_wells", "dt1_f64_cpr_gmres_with_wells", "dt20_f64_ilu0_with_wells", "dt20_f64_cpr_gmres_with_wells"} <= set(v)
    assert all("failed" not in x for x in v.values()), v
    assert "f32" in v["reference_default_solver_ilu0_with_wells"]["arithmetic"] and "reference's switch on the CURRENT step length" in v["dt30_f64_ilu0_with_wells"]["arithmetic"]
    kt = d["kernel_table"]
    assert "classes" in kt and "amg_vcycle" in kt["classes"] and kt["calls"] == 4 and kt["solving_iterations"] >= 1


def test_bench_distributed_path_with_one_rank(gpu_lib):
    d = _run("--force-dist", "--only-main", "--no-cpu-baseline")
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "1 GPU"
    assert d["config"]["linear_solver"].endswith("gmres(40)") and d["config"]["workload"].endswith("_fivespot")
    assert d["value"] == pytest.approx(d["config"]["cells"] / (d["ms_per_solving_iteration_median"] * 1e-3) / 1e6)
