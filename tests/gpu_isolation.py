"""Test infrastructure: every GPU test module runs in ONE fresh child process; the pytest process itself holds no HIP context.

Why: 295 GPU tests in one process under `-x` turn a single device fault -- HSA's
"memory access fault -> abort()" ends the whole interpreter -- into a red record that names neither the test nor the signal,
and everything after it counts as untested.  Here the parent only COLLECTS: when the run reaches the first selected test of
`tests/test_gpu_<m>.py` it starts `python -m pytest tests/test_gpu_<m>.py` as a child (a process that has never touched the
GPU starts a process that will), the child appends one JSON line per test phase to a results file (pytest's own report
serialisation, the one xdist uses), and the parent replays those reports under the tests' own node ids: `-x`, `-q`,
`--durations`, the pass / fail counts and the failure text are the child's, unchanged.

A child that dies (fatal signal, or the module's time limit):
  * the test it died in is reported FAILED with the signal's name, the child's exit status and the tail of the child's
    uncut output (native stderr included: the child runs with `--capture=sys`, so fd 2 -- HSA's message, the interpreter's
    fatal-signal dump of every thread -- goes straight into gpurun_out/gpu_suite/<module>.log);
  * the tests of the module that had not started yet run in another fresh child, so they get their own outcomes (at most
    DSP_GPU_MAX_DEATHS children die per module, default 3; the rest is then reported failed as "not run");
  * one line per death is appended to gpurun_out/gpu_suite/deaths.txt.
Nothing is retried: a test that killed its process has failed.

Switches: DSP_GPU_ISOLATE=0 runs everything in the pytest process as before (debuggers, `--pdb`);
DSP_GPU_MODULE_TIMEOUT seconds per child (default 1500, or the module's own GPU_MODULE_TIMEOUT attribute); DSP_GPU_CHILD_* are
set by the parent for the child.
"""
import json
import os
import signal
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT_DIR = os.path.join(ROOT, "gpurun_out", "gpu_suite")


def is_child():
    return bool(os.environ.get("DSP_GPU_CHILD_RESULTS"))


def isolating(config):
    if is_child() or os.environ.get("DSP_GPU_ISOLATE", "1") == "0":
        return False
    # a debugger needs the test in the process it is attached to
    return not (config.getoption("usepdb", False) or config.getoption("trace", False))


def is_gpu_item(item):
    return item.get_closest_marker("gpu") is not None


def _out_dir():
    d = os.environ.get("DSP_GPU_SUITE_DIR") or OUT_DIR
    os.makedirs(d, exist_ok=True)
    return d


# ------------------------------------------------------------------------------------------------------------ child side

class ChildRecorder:
    """In the child: keep only the node ids the parent selected, append every report to the results file as it happens
    (flushed and fsync'ed: a process that is killed in the next test must not take the earlier lines with it)."""

    def __init__(self, config):
        self.config = config
        self.path = os.environ["DSP_GPU_CHILD_RESULTS"]
        sel = os.environ.get("DSP_GPU_CHILD_SELECT")
        self.select = None
        if sel:
            with open(sel) as f:
                self.select = set(json.load(f))
        self.f = open(self.path, "a")

    def _put(self, obj):
        self.f.write(json.dumps(obj) + "\n")
        self.f.flush()
        os.fsync(self.f.fileno())

    @pytest.hookimpl(trylast=True)
    def pytest_collection_modifyitems(self, config, items):
        if self.select is None:
            return
        keep = [it for it in items if it.nodeid in self.select]
        drop = [it for it in items if it.nodeid not in self.select]
        if drop:
            config.hook.pytest_deselected(items=drop)
            items[:] = keep

    def pytest_runtest_logstart(self, nodeid, location):
        self._put({"kind": "start", "nodeid": nodeid, "t": time.time(), "pid": os.getpid()})

    def pytest_runtest_logreport(self, report):
        data = self.config.hook.pytest_report_to_serializable(config=self.config, report=report)
        self._put({"kind": "report", "nodeid": report.nodeid, "when": report.when, "data": data})

    def pytest_runtest_logfinish(self, nodeid, location):
        self._put({"kind": "finish", "nodeid": nodeid})


# ----------------------------------------------------------------------------------------------------------- parent side

def _read_results(path):
    out = []
    try:
        with open(path) as f:
            for line in f:
                line = line.strip()
                if not line:
                    continue
                try:
                    out.append(json.loads(line))
                except ValueError:
                    break   # the line the child was writing when it died
    except OSError:
        pass
    return out


def _tail(path, n_lines=80, max_bytes=24000):
    try:
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            size = f.tell()
            f.seek(max(0, size - max_bytes))
            text = f.read().decode("utf-8", "replace")
    except OSError:
        return "(no log)"
    return "\n".join(text.splitlines()[-n_lines:])


def _describe_exit(rc, timed_out):
    if timed_out:
        return "exceeded the module's time limit and was killed"
    if rc is not None and rc < 0:
        try:
            name = signal.Signals(-rc).name
        except ValueError:
            name = "signal %d" % -rc
        return "died of %s" % name
    return "exited with status %s before reporting" % rc


class ModuleRunner:
    """Runs the selected tests of one module in fresh children and hands back {nodeid: [report, ...]}."""

    def __init__(self, config, module_path, nodeids, limit=None):
        self.config = config
        self.module_path = module_path
        self.nodeids = list(nodeids)
        self.limit = limit     # the module's own GPU_MODULE_TIMEOUT (seconds), if it states one
        self.reports = {}      # nodeid -> list of deserialised reports
        self.synthetic = {}    # nodeid -> failure text (died / never ran)
        self.log_paths = []
        self.deaths = 0

    def _child_cmd(self):
        cmd = [sys.executable, "-m", "pytest", self.module_path, "-m", "gpu", "-p", "no:cacheprovider", "--capture=sys",
               "-v", "--tb=" + (self.config.getoption("tbstyle", "auto") or "auto"),
               "--rootdir", str(self.config.rootpath)]   # the child's node ids must be the parent's
        if self.config.inipath is not None:
            cmd += ["-c", str(self.config.inipath)]
        if self.config.getoption("maxfail", 0) == 1:
            cmd.append("-x")
        return cmd, str(self.config.rootpath)

    def run(self):
        stem = os.path.splitext(os.path.basename(self.module_path))[0]
        out_dir = _out_dir()
        remaining = list(self.nodeids)
        max_deaths = int(os.environ.get("DSP_GPU_MAX_DEATHS", "3"))
        limit = float(os.environ.get("DSP_GPU_MODULE_TIMEOUT") or self.limit or 1500)
        attempt = 0
        stop_after_failure = self.config.getoption("maxfail", 0) == 1
        while remaining:
            attempt += 1
            tag = stem if attempt == 1 else "%s.%d" % (stem, attempt)
            res_path = os.path.join(out_dir, tag + ".results.jsonl")
            sel_path = os.path.join(out_dir, tag + ".select.json")
            log_path = os.path.join(out_dir, tag + ".log")
            for p in (res_path, log_path):
                if os.path.exists(p):
                    os.remove(p)
            with open(sel_path, "w") as f:
                json.dump(remaining, f)
            env = dict(os.environ)
            env["DSP_GPU_CHILD_RESULTS"] = res_path
            env["DSP_GPU_CHILD_SELECT"] = sel_path
            env["PYTHONFAULTHANDLER"] = "1"
            env["PYTHONUNBUFFERED"] = "1"
            cmd, cwd = self._child_cmd()
            self.log_paths.append(log_path)
            timed_out = False
            with open(log_path, "wb") as log:
                log.write(("# %s\n" % " ".join(cmd)).encode())
                log.flush()
                proc = subprocess.Popen(cmd, cwd=cwd, env=env, stdin=subprocess.DEVNULL, stdout=log, stderr=subprocess.STDOUT,
                                        start_new_session=True)
                try:
                    rc = proc.wait(timeout=limit)
                except subprocess.TimeoutExpired:
                    timed_out = True
                    try:
                        os.killpg(proc.pid, signal.SIGKILL)   # the child's own process group: it and what it started
                    except OSError:
                        pass
                    rc = proc.wait()
            lines = _read_results(res_path)
            started, finished, failed_any = [], set(), False
            for ln in lines:
                nid = ln.get("nodeid")
                if ln["kind"] == "start":
                    started.append(nid)
                elif ln["kind"] == "report":
                    rep = self.config.hook.pytest_report_from_serializable(config=self.config, data=ln["data"])
                    # JSON has no tuples; pytest's terminal reporter insists on them (xdist's channel keeps them)
                    if isinstance(getattr(rep, "longrepr", None), list):
                        rep.longrepr = tuple(rep.longrepr)
                    if isinstance(getattr(rep, "location", None), list):
                        rep.location = tuple(rep.location)
                    self.reports.setdefault(nid, []).append(rep)
                    failed_any = failed_any or rep.failed
                elif ln["kind"] == "finish":
                    finished.add(nid)
            done = [n for n in remaining if n in finished]
            not_done = [n for n in remaining if n not in finished]
            if not not_done:
                break
            # pytest's own exit codes 0 / 1 with every started test finished: the child stopped on purpose (-x after a
            # failure, or a deselection inside the child).  Anything else with unfinished tests is a death.
            in_flight = [n for n in started if n not in finished]
            clean = (not timed_out) and rc in (0, 1, 5) and not in_flight
            if clean:
                if failed_any and stop_after_failure:
                    break   # the parent's -x ends the session at that failure; the rest is rightly never reported
                for n in not_done:
                    self.synthetic[n] = ("the child process of this module (%s) finished with status %s without running this "
                                         "test\n%s" % (" ".join(cmd), rc, _tail(log_path)))
                break
            self.deaths += 1
            how = _describe_exit(rc, timed_out)
            victim = in_flight[0] if in_flight else None
            with open(os.path.join(out_dir, "deaths.txt"), "a") as f:
                f.write("%.1f %s: child %s in %s (log %s)\n" % (time.time(), stem, how, victim or "(between tests)", log_path))
            text = ("the child process running this module %s%s\n  command: %s\n  full output: %s\n"
                    "---- last lines of the child's output (stdout + native stderr) ----\n%s"
                    % (how, " while running this test" if victim else "", " ".join(cmd), log_path, _tail(log_path)))
            if victim is not None:
                self.reports.pop(victim, None)   # a setup report without its call must not be replayed as a pass
                self.synthetic[victim] = text
                not_done = [n for n in not_done if n != victim]
            elif not done:
                # died before its first test started (collection, import): running it again would only repeat that
                for n in not_done:
                    self.synthetic[n] = text
                break
            if stop_after_failure:
                break
            if self.deaths >= max_deaths:
                for n in not_done:
                    self.synthetic[n] = "not run: %d child processes of this module died already\n%s" % (self.deaths, text)
                break
            remaining = not_done
        return self


class GpuIsolation:
    """Parent-side plugin: replaces the run protocol of every gpu-marked item by the replay of its child's reports."""

    def __init__(self, config):
        self.config = config
        self.by_module = {}   # module path -> ordered node ids selected in this session
        self.limits = {}      # module path -> its GPU_MODULE_TIMEOUT attribute
        self.runners = {}

    @pytest.hookimpl(trylast=True)
    def pytest_collection_modifyitems(self, config, items):
        self.by_module = {}
        for it in items:
            if is_gpu_item(it):
                self.by_module.setdefault(str(it.path), []).append(it.nodeid)
                self.limits[str(it.path)] = getattr(getattr(it, "module", None), "GPU_MODULE_TIMEOUT", None)
        if self.by_module:
            # The collecting process maps the product library too (dlopen + the pure dsp_abi_version() of _native.lib(): no HIP
            # call, no device touched -- the tests still run in the children): whoever audits "which in-tree .so did the pytest
            # process load" by looking at THIS process finds the library the children compute with, not an empty list.
            try:
                from deepsignal_plant_amd import _native
                _native.lib()
            except Exception:   # (no library built: the children's tests say so themselves, loudly)
                pass

    @pytest.hookimpl(tryfirst=True)
    def pytest_runtest_protocol(self, item, nextitem):
        if not is_gpu_item(item):
            return None
        mod = str(item.path)
        runner = self.runners.get(mod)
        if runner is None:
            runner = self.runners[mod] = ModuleRunner(self.config, mod, self.by_module.get(mod, [item.nodeid]), self.limits.get(mod)).run()
        ihook = item.ihook
        ihook.pytest_runtest_logstart(nodeid=item.nodeid, location=item.location)
        reports = runner.reports.get(item.nodeid)
        if item.nodeid in runner.synthetic or not reports:
            text = runner.synthetic.get(item.nodeid, "the child process of this module reported nothing for this test")
            rep = pytest.TestReport(nodeid=item.nodeid, location=item.location, keywords={k: 1 for k in item.keywords},
                                    outcome="failed", longrepr=text, when="call", sections=[], duration=0.0)
            ihook.pytest_runtest_logreport(report=rep)
        else:
            for rep in reports:
                ihook.pytest_runtest_logreport(report=rep)
        ihook.pytest_runtest_logfinish(nodeid=item.nodeid, location=item.location)
        return True

    def pytest_terminal_summary(self, terminalreporter):
        deaths = sum(r.deaths for r in self.runners.values())
        if deaths:
            terminalreporter.write_line("GPU suite: %d child process(es) died -- %s" % (deaths, os.path.join(_out_dir(), "deaths.txt")),
                                        red=True)


def configure(config):
    """Called from tests/conftest.py:pytest_configure."""
    if is_child():
        config.pluginmanager.register(ChildRecorder(config), "dsp-gpu-child")
    elif isolating(config):
        config.pluginmanager.register(GpuIsolation(config), "dsp-gpu-isolation")
