"""bench.py's rank launcher (no GPU needed): `--gpus N` without a launcher becomes a torch.distributed.run command line with N
ranks on 127.0.0.1; under a launcher the world it made must be the one asked for."""
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_gpus_flag_builds_the_launcher_command(monkeypatch):
    bench = _bench()
    seen = {}

    def fake_execv(exe, argv):
        seen["exe"], seen["argv"] = exe, list(argv)
        raise SystemExit(0)
    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    with pytest.raises(SystemExit):
        bench.launch_ranks_if_needed(types.SimpleNamespace(gpus=4))
    a = seen["argv"]
    assert a[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in a
    assert a[a.index("--nproc-per-node") + 1] == "4" and a[a.index("--master-addr") + 1] == "127.0.0.1"
    assert a[-5].endswith("bench.py") and a[-4:] == ["--gpus", "4", "--steps", "3"]


def test_single_gpu_and_launched_ranks_do_not_relaunch(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(os, "execv", lambda *a: (_ for _ in ()).throw(AssertionError("must not exec")))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    bench.launch_ranks_if_needed(types.SimpleNamespace(gpus=1))
    monkeypatch.setenv("WORLD_SIZE", "8")
    bench.launch_ranks_if_needed(types.SimpleNamespace(gpus=8))
    with pytest.raises(SystemExit):
        bench.launch_ranks_if_needed(types.SimpleNamespace(gpus=4))      # a world of 8 was asked to report as 4
