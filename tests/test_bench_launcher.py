"""`python bench.py --gpus N` without a launcher starts its own N ranks (one process per GPU) before anything
touches the GPU, forwards rank 0's single JSON line and exits with the worst rank's code."""
import importlib.util
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BENCH = os.path.join(ROOT, "bench.py")


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_rank_environment():
    b = _bench_module()
    env = b.rank_environment({"PATH": "/x", "RANK": "9"}, 3, 8, 4711)
    assert env["RANK"] == "3" and env["LOCAL_RANK"] == "3" and env["WORLD_SIZE"] == "8" and env["LOCAL_WORLD_SIZE"] == "8"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "4711" and env["PATH"] == "/x"
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert b.rank_environment({"HSA_ENABLE_IPC_MODE_LEGACY": "1"}, 0, 2, 1)["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"


def _run(args, extra_env=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_launcher_starts_every_rank_and_forwards_rank0_only():
    r = _run(["--gpus", "4", "--steps", "7"], {"ORBHIP_BENCH_LAUNCH_SELFTEST": "1"})
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # rank 0's line alone on stdout
    j = json.loads(lines[0])
    assert j == {"rank": 0, "local_rank": 0, "world": 4, "master": "127.0.0.1", "port": j["port"], "steps": 7}
    others = [json.loads(l) for l in r.stderr.splitlines() if l.startswith("{")]
    assert sorted(o["rank"] for o in others) == [1, 2, 3]
    assert all(o["world"] == 4 and o["port"] == j["port"] for o in others)


def test_launcher_returns_the_worst_exit_code():
    r = _run(["--gpus", "2"], {"ORBHIP_BENCH_LAUNCH_SELFTEST": "1", "ORBHIP_BENCH_LAUNCH_SELFTEST_RC_RANK1": "3"})
    assert r.returncode == 3
    assert "exit codes of the 2 ranks: [0, 3]" in r.stderr


def test_under_a_launcher_no_ranks_are_started():
    # WORLD_SIZE set (torch.distributed.run did the launching): the process is a rank, not a launcher
    r = _run(["--gpus", "2"], {"ORBHIP_BENCH_LAUNCH_SELFTEST": "1", "RANK": "1", "LOCAL_RANK": "1", "WORLD_SIZE": "2"})
    assert r.returncode == 0
    assert json.loads(r.stdout.strip())["rank"] == 1


def test_without_a_gpu_every_rank_fails_cleanly():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present")
    r = _run(["--gpus", "2", "--steps", "2", "--no-extra"])
    assert r.returncode != 0
    assert r.stderr.count("needs a GPU") == 2
    assert "torch.distributed.run" not in r.stderr
    assert r.stdout.strip() == ""


class _NoCollectives:
    """Stands in for torch.distributed in a one-rank call of run_dist_legs_isolated."""

    @staticmethod
    def broadcast_object_list(objs, src=0):
        return None


def _isolated(monkeypatch, hook, timeout=60):
    import argparse
    import bench
    monkeypatch.setenv("ORBHIP_BENCH_LEGS_CHILD_SELFTEST", hook)
    args = argparse.Namespace(backend="gloo", legs_timeout=timeout)
    return bench.run_dist_legs_isolated(args, _NoCollectives, 0, 0, 1)


def test_isolated_legs_return_what_the_child_prints(monkeypatch):
    legs = _isolated(monkeypatch, "answer")
    assert legs["selftest"] == "answer" and legs["rank"] == 0 and legs["world"] == 1 and int(legs["port"]) > 0


def test_isolated_legs_survive_a_child_that_aborts(monkeypatch):
    legs = _isolated(monkeypatch, "abort")
    assert "error" in legs and "exit code" in legs["error"]


def test_isolated_legs_survive_a_child_that_hangs(monkeypatch):
    legs = _isolated(monkeypatch, "hang", timeout=3)
    assert "error" in legs and "did not finish" in legs["error"]
