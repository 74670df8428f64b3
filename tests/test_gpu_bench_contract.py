"""GPU: `python bench.py` on a small deck prints ONE JSON line of < 4 KB that carries the driver's contract (metric / value / unit / n_gpus /
steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus the `roofline`, `cpu_baseline`
and `per_time_step` objects and the compact variant map; the full records live in the detail file; the distributed code path with one
rank prints the same contract."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, detail="bench_detail_test.json"):
    dpath = os.path.join(ROOT, "gpurun_out", detail)
    os.makedirs(os.path.dirname(dpath), exist_ok=True)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--nx", "24", "--ny", "24", "--nz", "12", "--steps", "4", "--warmup", "1", "--rate", "10",
                          "--stat-calls", "24", "--detail", dpath, *args], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]          # stdout carries the ONE line and nothing else
    # the driver keeps 8 KB of stdout -- the line stays under 4 KB, everything else lives in the detail file
    assert len(lines[0]) < 4096, len(lines[0])
    with open(dpath) as f:
        return json.loads(lines[0]), json.load(f)


def test_bench_line_contract(gpu_lib):
    d, full = _run()
    assert d["metric"].startswith("Mcell-updates/sec") and d["unit"] == "Mcell-updates/s"
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f64"
    # value = cells / MEDIAN duration of the timed iterations that include a solve; the mean over the solving calls and over all calls beside it
    assert d["value"] > 0 and abs(d["value"] - d["config"]["cells"] / (d["ms_per_solving_iteration_median"] * 1e-3) / 1e6) < 2e-3 * d["value"]
    assert abs(d["value_mean_solving"] - d["config"]["cells"] / (d["ms_per_solving_iteration_mean"] * 1e-3) / 1e6) < 2e-3 * d["value_mean_solving"]
    assert abs(d["value_all_calls_mean"] - d["config"]["cells"] / (d["ms_per_step"] * 1e-3) / 1e6) < 2e-3 * d["value_all_calls_mean"]
    main = full["main"]
    calls = main["calls"]
    assert calls["timed"] == 4 and len(calls["ms"]) == 24 and sum(calls["solved"][:4]) == d["config"]["solving_iterations"] >= 1
    import statistics
    assert abs(statistics.median(m for m, s in zip(calls["ms"][:4], calls["solved"][:4]) if s) - d["ms_per_solving_iteration_median"]) < 2e-3
    assert sum(calls["ms"][:4]) <= 1.05 * d["ms_per_step"] * 4 + 0.5           # the per-call durations add up to the timed region
    b = d["breakdown_ms"]
    assert b["assemble"] + b["linear_solve"] + b["update"] <= d["ms_per_solving_iteration_median"] * 1.02 + 3e-3
    # per time step: whole converged steps inside the 24 calls
    p = d["per_time_step"]
    assert p["time_steps"] >= 2 and 1.0 <= p["newton_iterations_per_time_step"] <= 10.0 and p["ms_per_converged_time_step"] > 0 and p["ms_per_simulated_day"] > 0
    ev = calls["event"]
    assert ev.count("step") >= p["time_steps"] and abs(p["simulated_days"] - 5.0 * p["time_steps"]) < 1e-9 + 5.0 * p["chopped_attempts"] * p["time_steps"]
    # the headline: CPR (cpr_use_amg, one V-cycle) in double (NewtonIterationBlackoilCPR.cpp:117-140) with newton_use_gmres, stage 2 relaxed by cpr_relax = 1
    c = d["config"]
    assert c["linear_solver"] == "cpr(amg V-cycle + ilu0, relax 1) + gmres(40)" and full["gmres_true_residual_check"] is False
    assert "cpr_use_amg=true" in c["reference_equivalence"] and "newton_use_gmres=true" in c["reference_equivalence"] and "DIFFERS" in c["reference_equivalence"]
    assert c["workload"].startswith("cart24x24x12") and c["workload"].endswith("_fivespot") and "model" not in c and len(c) <= 16
    assert c["time_steps_not_converged"] == 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["traffic"] is None
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["kernel"].startswith("k_spmv")
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == d["unit"] and cb["sample"]
    v = d["variants"]
    names = {"ref_default_ilu0", "ilu1_bicgstab", "cpr_f64_gmres_ilu1", "cpr_ref_defaults", "cpr_amg_inner_bicgstab", "without_wells", "cpr_f64_bicgstab", "cpr_f64_gmres_verified", "cpr_f64_gmres_fixed_factor",
             "cpr_f32_gmres", "cpr_f64_gmres_f32_precond", "cpr_f64_bicgstab_f32_precond", "cpr_f64_bicgstab_damped", "dt30_f64_ilu0", "dt30_f64_cpr_gmres", "dt30_f64_cpr_bicgstab", "dt1_f32_ilu0", "dt1_f64_cpr_gmres", "dt20_f64_ilu0", "dt20_f64_cpr_gmres"}
    assert names <= set(v), names - set(v)
    assert all(x != "failed" and x[0] > 0 for x in v.values()), v
    fv = full["same_run_variants"]
    assert set(fv) == set(v) and all("failed" not in x for x in fv.values())
    assert "f32" in fv["ref_default_ilu0"]["arithmetic"] and "reference's switch on the CURRENT step length" in fv["dt30_f64_ilu0"]["arithmetic"]
    kt = full["kernel_table"]
    assert "classes" in kt and "amg_vcycle" in kt["classes"] and kt["calls"] == 4 and kt["solving_iterations"] >= 1
    assert d["decks"] == {}             # the other BASELINE decks ride on the full-size run only


def test_bench_distributed_path_with_one_rank(gpu_lib):
    d, _ = _run("--force-dist", "--only-main", "--no-cpu-baseline", detail="bench_detail_test_dist.json")
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "1 GPU"
    assert d["config"]["linear_solver"].endswith("gmres(40)") and d["config"]["workload"].endswith("_fivespot")
    assert d["value"] == pytest.approx(d["config"]["cells"] / (d["ms_per_solving_iteration_median"] * 1e-3) / 1e6, rel=2e-3)


def test_bench_norne_like_deck_decomposed_over_the_test_transport(gpu_lib):
    """BASELINE configs[4] decomposed: `bench.py --deck nornelike --gpus 2` launched like the driver launches it (torch.distributed.run, one rank
    per process; here both ranks on cuda:0, coupled by the shared-memory TEST transport and gloo): the ACTNUM deck is cut into slabs of whole
    j-rows balanced by active cells, every vertical well stays on one rank, the run prints the contract line and chops no time step."""
    dpath = os.path.join(ROOT, "gpurun_out", "bench_detail_test_norne2.json")
    env = dict(os.environ, OPMGPU_COMM_TRANSPORT="shm")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29977",
                          os.path.join(ROOT, "bench.py"), "--gpus", "2", "--deck", "nornelike", "--steps", "30", "--warmup", "2", "--stat-calls", "30", "--no-cpu-baseline",
                          "--detail", dpath], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["workload"] == "nornelike_46x112x22_3phase_blackoil_36wells"
    assert 0.4 * d["config"]["cells"] < d["config"]["cells_per_gpu"] < 0.6 * d["config"]["cells"]
    assert d["per_time_step"]["chopped_attempts"] == 0 and d["per_time_step"]["time_steps"] >= 5, d["per_time_step"]
    assert d["config"]["linear_its_per_solve"] < 20, d["config"]
