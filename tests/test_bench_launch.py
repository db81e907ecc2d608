"""CPU: `python bench.py --gpus N` without a launcher starts its own ranks (a fresh torch.distributed.run child) and
passes their exit code on; the accounting helper of `roofline.path_frac` matches SURVEY 8(d)."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_self_launch_command(monkeypatch):
    b = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    args = b.parse_args()
    assert b.self_launch(args) == 7  # the child's exit code is the parent's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def test_gpus_2_without_launcher_spawns_ranks_and_fails_loudly_without_a_gpu():
    """end to end on the CPU box: the parent launches two ranks, each refuses to run without an MI355X (there is
    no CPU fallback), and the failure comes back as a non-zero exit code"""
    try:
        import torch
        if torch.cuda.is_available():
            import pytest
            pytest.skip("only meaningful on a box without a GPU")
    except ImportError:
        pass
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--bytes", "1000000", "--no-cpu", "--no-extra"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert "needs an MI355X" in (p.stderr + p.stdout)


def test_path_accounting_matches_survey_8d():
    b = _bench()
    st = {"raw_bytes": 1000, "rle_bytes": 900, "bwt_active_sum": 2000, "mtf_syms": 400, "out_bits": 8 * 250}
    assert b.path_alg_bytes(st) == 1000 + 97 * 900 + 96 * 2000 + 6 * 400 + 3 * 250
