"""CPU: the scripts under tools/ still load (VERDICT r2 hygiene: 37 files with nothing running them).  Every Python
script byte-compiles; every script that takes its arguments through argparse answers --help (which imports what it
imports, on this GPU-less host, and touches no GPU); every shell script parses; every .hip micro-benchmark compiles for
gfx950 when asked to (DSP_TEST_MICRO=1: about 20 s each, so not by default)."""
import glob
import os
import py_compile
import subprocess
import sys

import pytest

from tests.helpers import ROOT

TOOLS = os.path.join(ROOT, "tools")
PY = sorted(glob.glob(os.path.join(TOOLS, "*.py")) + glob.glob(os.path.join(TOOLS, "*", "*.py")))
SH = sorted(glob.glob(os.path.join(TOOLS, "*.sh")) + glob.glob(os.path.join(TOOLS, "*", "*.sh")))
HIP = sorted(glob.glob(os.path.join(TOOLS, "micro", "*.hip")))


def test_there_are_tools_to_check():
    assert len(PY) >= 20 and len(SH) >= 3 and len(HIP) >= 5


@pytest.mark.parametrize("path", PY, ids=[os.path.relpath(p, TOOLS) for p in PY])
def test_python_tool_compiles_and_answers_help(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "x.pyc"), doraise=True)
    src = open(path).read()
    if "argparse" not in src or "parse_args" not in src:
        return
    env = dict(os.environ, DSP_WORK=str(tmp_path))
    r = subprocess.run([sys.executable, path, "--help"], cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "usage" in r.stdout.lower(), (r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("path", SH, ids=[os.path.relpath(p, TOOLS) for p in SH])
def test_shell_tool_parses(path):
    r = subprocess.run(["bash", "-n", path], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr


@pytest.mark.skipif(not os.environ.get("DSP_TEST_MICRO"), reason="set DSP_TEST_MICRO=1 to cross-compile every micro-benchmark")
@pytest.mark.parametrize("path", HIP, ids=[os.path.basename(p) for p in HIP])
def test_micro_benchmark_compiles_for_gfx950(path, tmp_path):
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-w", "-o", str(tmp_path / "m"), path],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
