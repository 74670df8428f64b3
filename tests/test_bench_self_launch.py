"""CPU: bench.py --gpus N launched without torch.distributed.run starts the launcher itself as a child process."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launch_builds_the_drivers_command(monkeypatch):
    """`python bench.py --gpus N` without WORLD_SIZE starts torch.distributed.run as a CHILD process before anything touches the GPU and
    relays the child's single JSON line (the child is mocked here; no GPU needed)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class P:
        returncode = 0
        stdout = 'noise\n{"metric": "m", "value": 1.0}\n'

    def fake_run(cmd, **kw):
        seen["cmd"], seen["env"] = cmd, kw["env"]
        return P()

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setitem(sys.modules, "torch", None)            # importing torch in the parent would be the bug
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    P.returncode = 3
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2"])
    assert e.value.code == 3


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_per_time_step_statistics_count_whole_steps_only():
    """bench.per_time_step_stats: the window runs from the call after the first step boundary to the last completed step;
    chopped attempts inside it count towards the cost of the steps, calls before / after it do not."""
    bench = _load_bench()
    day = 86400.0
    #          0       1        2       3        4       5        6       7        8       9
    log = [(None, 5 * day), ("step", 5 * day), (None, 5 * day), (None, 5 * day), ("chop", 5 * day), (None, 1.65 * day), ("step", 1.65 * day),
           (None, 3.3 * day), ("step", 3.3 * day), (None, 5 * day)]
    ms = [1.0, 0.5, 2.0, 2.0, 2.0, 2.0, 0.5, 2.0, 0.5, 2.0]
    solved = [1, 0, 1, 1, 1, 1, 0, 1, 0, 1]
    lin = [3, 0, 4, 4, 5, 3, 0, 3, 0, 3]
    p = bench.per_time_step_stats(log, ms, solved, lin)
    assert p["time_steps"] == 2 and p["calls"] == 7 and p["chopped_attempts"] == 1          # calls 2..8
    assert p["newton_iterations_per_time_step"] == 5 / 2 and p["linear_iterations_per_time_step"] == 19 / 2
    assert abs(p["ms_per_converged_time_step"] - 11.0 / 2) < 1e-12
    assert abs(p["simulated_days"] - 4.95) < 1e-12 and abs(p["ms_per_simulated_day"] - 11.0 / 4.95) < 1e-12
    # no completed step behind the first boundary: nothing to report
    assert bench.per_time_step_stats(log[:3], ms[:3], solved[:3], lin[:3]) is None
    assert bench.per_time_step_stats([(None, day)] * 4, [1.0] * 4, [1] * 4, [1] * 4) is None


def test_bench_line_budget_constant():
    assert _load_bench().MAX_LINE_BYTES == 4096
