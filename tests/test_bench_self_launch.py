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
