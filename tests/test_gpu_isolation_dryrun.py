"""CPU dry run of tests/gpu_isolation.py: three fake `gpu` modules -- no GPU involved -- run through the
same plugin; the middle one kills its own process with SIGABRT in its second test, as HSA's abort() after a memory access
fault would.  Exactly that test fails, with the child's native stderr attached; every other test -- the ones of the same
module that had not started included -- gets its own outcome from a fresh child."""
import os
import re
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFTEST = """
import os, sys
sys.path.insert(0, %r)
from tests import gpu_isolation

def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: fake")
    gpu_isolation.configure(config)
""" % ROOT

MOD_A = """
import os, pytest
pytestmark = pytest.mark.gpu

def test_a1():
    assert os.environ.get("DSP_GPU_CHILD_RESULTS"), "must run in a child"

@pytest.mark.parametrize("n", [1, 2])
def test_a2(n):
    print("a2 says", n)

def test_a_skip():
    pytest.skip("nothing to do here")
"""

MOD_B = """
import os, signal, sys, pytest
pytestmark = pytest.mark.gpu

def test_b1():
    pass

def test_b2_dies():
    os.write(2, b"HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION: pretend message on fd 2\\n")
    os.kill(os.getpid(), signal.SIGABRT)

def test_b3():
    pass

def test_b4_fails():
    assert 1 + 1 == 3, "an ordinary failure"
"""

MOD_C = """
import pytest
pytestmark = pytest.mark.gpu

def test_c1():
    pass
"""

MOD_CPU = """
import os
def test_cpu_runs_in_the_parent():
    assert not os.environ.get("DSP_GPU_CHILD_RESULTS")
"""


def _tree(tmp_path):
    (tmp_path / "conftest.py").write_text(CONFTEST)
    (tmp_path / "test_gpu_a.py").write_text(textwrap.dedent(MOD_A))
    (tmp_path / "test_gpu_b.py").write_text(textwrap.dedent(MOD_B))
    (tmp_path / "test_gpu_c.py").write_text(textwrap.dedent(MOD_C))
    (tmp_path / "test_plain.py").write_text(textwrap.dedent(MOD_CPU))


def _run(tmp_path, *extra):
    env = {k: v for k, v in os.environ.items() if not k.startswith("DSP_GPU_")}
    env["DSP_GPU_SUITE_DIR"] = str(tmp_path / "suite_out")
    return subprocess.run([sys.executable, "-m", "pytest", str(tmp_path), "-p", "no:cacheprovider", "--rootdir", str(tmp_path),
                           "-rA"] + list(extra), cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)


def _outcomes(stdout):
    return {m.group(2): m.group(1) for m in re.finditer(r"^(PASSED|FAILED|SKIPPED|ERROR) (\S+)", stdout, re.M)}


def test_a_child_that_dies_costs_exactly_the_test_it_died_in(tmp_path):
    _tree(tmp_path)
    r = _run(tmp_path, "-v")
    out = r.stdout + r.stderr
    got = _outcomes(r.stdout)
    assert got.get("test_gpu_a.py::test_a1") == "PASSED", out
    assert got.get("test_gpu_a.py::test_a2[1]") == "PASSED" and got.get("test_gpu_a.py::test_a2[2]") == "PASSED", out
    assert got.get("test_gpu_b.py::test_b1") == "PASSED", out
    assert got.get("test_gpu_b.py::test_b2_dies") == "FAILED", out
    assert got.get("test_gpu_b.py::test_b3") == "PASSED", out          # ran in a second, fresh child
    assert got.get("test_gpu_b.py::test_b4_fails") == "FAILED", out    # an ordinary failure keeps its own text
    assert got.get("test_gpu_c.py::test_c1") == "PASSED", out
    assert got.get("test_plain.py::test_cpu_runs_in_the_parent") == "PASSED", out
    assert "1 skipped" in r.stdout and "nothing to do here" in r.stdout, out
    assert "died of SIGABRT while running this test" in r.stdout, out
    assert "HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION: pretend message on fd 2" in r.stdout, out   # native stderr attached
    assert "an ordinary failure" in r.stdout, out
    assert "2 failed, 7 passed, 1 skipped" in r.stdout, out
    assert r.returncode == 1
    deaths = (tmp_path / "suite_out" / "deaths.txt").read_text()
    assert deaths.count("\n") == 1 and "test_gpu_b.py::test_b2_dies" in deaths and "SIGABRT" in deaths
    log = (tmp_path / "suite_out" / "test_gpu_b.log").read_text()
    assert "Fatal Python error: Aborted" in log      # the interpreter's dump of every thread, uncut, in the module's log


def test_with_x_the_session_stops_at_the_death_and_names_it(tmp_path):
    _tree(tmp_path)
    r = _run(tmp_path, "-x", "-q", "-m", "gpu")
    out = r.stdout + r.stderr
    got = _outcomes(r.stdout)
    assert got.get("test_gpu_b.py::test_b2_dies") == "FAILED", out
    assert "test_gpu_b.py::test_b3" not in got and "test_gpu_c.py::test_c1" not in got, out
    assert "1 failed, 4 passed, 1 skipped" in r.stdout and "1 deselected" in r.stdout, out


def test_a_selection_in_the_parent_is_the_selection_in_the_child(tmp_path):
    _tree(tmp_path)
    r = _run(tmp_path, "-q", "-m", "gpu", "-k", "a2 or c1")
    got = _outcomes(r.stdout)
    assert sorted(got) == ["test_gpu_a.py::test_a2[1]", "test_gpu_a.py::test_a2[2]", "test_gpu_c.py::test_c1"], r.stdout
    assert "3 passed" in r.stdout and r.returncode == 0, r.stdout


def test_a_module_over_its_time_limit_is_killed_and_named(tmp_path):
    _tree(tmp_path)
    (tmp_path / "test_gpu_b.py").write_text(textwrap.dedent("""
        import time, pytest
        pytestmark = pytest.mark.gpu
        def test_hangs():
            time.sleep(600)
        def test_after():
            pass
    """))
    env_limit = {"DSP_GPU_MODULE_TIMEOUT": "3"}
    os.environ.update(env_limit)
    try:
        env = {k: v for k, v in os.environ.items() if not k.startswith("DSP_GPU_") or k == "DSP_GPU_MODULE_TIMEOUT"}
        env["DSP_GPU_SUITE_DIR"] = str(tmp_path / "suite_out")
        r = subprocess.run([sys.executable, "-m", "pytest", str(tmp_path), "-p", "no:cacheprovider", "--rootdir", str(tmp_path),
                            "-rA", "-m", "gpu", "-k", "gpu_b"], cwd=str(tmp_path), env=env, capture_output=True, text=True,
                           timeout=300)
    finally:
        os.environ.pop("DSP_GPU_MODULE_TIMEOUT", None)
    got = _outcomes(r.stdout)
    assert got.get("test_gpu_b.py::test_hangs") == "FAILED" and got.get("test_gpu_b.py::test_after") == "PASSED", r.stdout
    assert "exceeded the module's time limit" in r.stdout, r.stdout


def test_a_module_states_its_own_time_limit(tmp_path):
    """GPU_MODULE_TIMEOUT in the module (tests/test_gpu_zz_extents.py raises the default for its sweeps): used when the
    environment does not set DSP_GPU_MODULE_TIMEOUT"""
    _tree(tmp_path)
    (tmp_path / "test_gpu_b.py").write_text(textwrap.dedent("""
        import time, pytest
        pytestmark = pytest.mark.gpu
        GPU_MODULE_TIMEOUT = 2
        def test_hangs():
            time.sleep(600)
        def test_after():
            pass
    """))
    r = _run(tmp_path, "-rA", "-m", "gpu", "-k", "gpu_b")
    got = _outcomes(r.stdout)
    assert got.get("test_gpu_b.py::test_hangs") == "FAILED" and got.get("test_gpu_b.py::test_after") == "PASSED", r.stdout
    assert "exceeded the module's time limit" in r.stdout, r.stdout
