"""bnzhip, the `bnz`-compatible CLI over libbzhip.so (reference bnz/src/main.rs).  Argument grammar,
exit codes and the keep/remove policy on CPU; end-to-end output bit-exact vs the oracle on the GPU."""
import os
import subprocess

import pytest

from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "banzai_amd", "bnzhip")


@pytest.fixture(scope="module")
def cli(native):
    assert os.path.exists(BIN), "build with `make -C banzai_amd/csrc`"
    return BIN


def run(cli, *args, stdin=None):
    return subprocess.run([cli, *args], input=stdin, capture_output=True)


def test_exit_codes_for_argument_errors(cli, tmp_path):
    """bnz/src/main.rs:11-14: 0 ok, 1 arguments, 2 filesystem, 3 output"""
    assert run(cli).returncode == 1                       # synopsis
    assert run(cli, "--bogus").returncode == 1
    assert run(cli, "-x", "f").returncode == 1            # invalid short flag
    assert run(cli, "a", "b").returncode == 1             # two inputs
    assert run(cli, "--output").returncode == 1           # no input
    assert run(cli, "--output", "-k", "f").returncode == 1  # --output needs a path
    assert run(cli, "-c", "--output", "x", "f").returncode == 1  # two outputs
    assert run(cli, str(tmp_path / "missing")).returncode == 2
    for flag in ("--help", "--info", "--version"):
        r = run(cli, flag)
        assert r.returncode == 0 and r.stderr


@pytest.mark.gpu
def test_cli_end_to_end(cli, oracle, tmp_path):
    d = cases.gen(300_000, "text", 1) + cases.gen(50_000, "longruns", 1)
    # default: writes <input>.bz2 at level 9 and removes the input
    f = tmp_path / "a.txt"
    f.write_bytes(d)
    assert run(cli, str(f)).returncode == 0
    assert not f.exists() and (tmp_path / "a.txt.bz2").read_bytes() == oracle.encode(d, 9)
    # explicit output keeps the input; -1 sets the level; -r removes anyway
    f.write_bytes(d)
    out = tmp_path / "o.bz2"
    assert run(cli, "-1", "--output", str(out), str(f)).returncode == 0
    assert f.exists() and out.read_bytes() == oracle.encode(d, 1)
    assert run(cli, "-r", "--fast", "--output", str(out), str(f)).returncode == 0
    assert not f.exists()
    # stdin -> stdout, combined short flags
    r = run(cli, "-c5", "-", stdin=d)
    assert r.returncode == 0 and r.stdout == oracle.encode(d, 5)
    # -k with default output name
    f.write_bytes(d)
    assert run(cli, "-k", str(f)).returncode == 0 and f.exists()
    # an input of several 16 MiB reads (the CLI feeds the streaming API), through a pipe
    big = cases.gen(20_000_000, "shortruns", 3) + cases.gen(17_000_000, "text", 3)
    r = run(cli, "-c", "-", stdin=big)
    assert r.returncode == 0 and r.stdout == oracle.encode(big, 9)
    # BZHIP_HUFFMAN=fixed: the opt-in Huffman mode -- a smaller, still valid stream
    import bz2
    import subprocess
    env = dict(os.environ, BZHIP_HUFFMAN="fixed")
    r2 = subprocess.run([cli, "-c", "-"], input=big, capture_output=True, env=env)
    assert r2.returncode == 0 and len(r2.stdout) < len(r.stdout) and bz2.decompress(r2.stdout) == big
    # BZHIP_DEVICES: the blocks spread over several GPUs of the node (here: the one GPU, listed three times) -- the same bytes
    env = dict(os.environ, BZHIP_DEVICES="0,0,0")
    r3 = subprocess.run([cli, "-c", "-"], input=big, capture_output=True, env=env)
    assert r3.returncode == 0 and r3.stdout == r.stdout
    f.write_bytes(d)
    r4 = subprocess.run([cli, "-5", "--output", str(out), str(f)], capture_output=True, env=env)
    assert r4.returncode == 0 and f.exists() and out.read_bytes() == oracle.encode(d, 5)
