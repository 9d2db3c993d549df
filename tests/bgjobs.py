"""Test infrastructure: the suite's long native runs (sanitizer builds of the host code, the protocol model, the parser harness --
each "build a binary, run it, read its output") start when collection ends and run NEXT TO the Python-level tests, four at a
time, instead of one after another in front of them: the container has 8 cores and pytest uses one.  A test declares its jobs,

    @bgjobs.uses(lambda params: ["host_asan_" + params["sanitizers"]])
    def test_...(sanitizers):
        r = bgjobs.result("host_asan_" + sanitizers)      # CompletedProcess (joins the job; starts it now if nobody did)

and a job is a function () -> subprocess.CompletedProcess registered under a name (bgjobs.job).  Only the jobs of SELECTED tests
are started (-k, -m, a single file: nothing else runs); a job's exception is re-raised in the test that asks for its result.
DSP_TEST_BGJOBS=0: every job runs inline, when its test asks for it."""
import concurrent.futures
import os
import threading

_REGISTRY = {}        # name -> function
_MODULE_JOBS = {}     # test file's basename -> job names every selected test of that file wants (module fixtures)
_FUTURES = {}         # name -> Future
_LOCK = threading.Lock()
_POOL = None


def job(name):
    def deco(fn):
        _REGISTRY[name] = fn
        return fn
    return deco


def uses(names_of):
    """names_of(params) -> the job names the test will ask for (params: the parametrisation of the item, {} if none)"""
    def deco(fn):
        fn._bgjobs = names_of
        return fn
    return deco


def module_uses(basename, names):
    """any selected test of the file `basename` starts these jobs (what the file's module-scoped fixtures wait for)"""
    _MODULE_JOBS[basename] = list(names)


def _enabled():
    return os.environ.get("DSP_TEST_BGJOBS", "1") != "0"


def start(name):
    global _POOL
    with _LOCK:
        if name in _FUTURES or name not in _REGISTRY:
            return
        if _POOL is None:
            _POOL = concurrent.futures.ThreadPoolExecutor(max_workers=int(os.environ.get("DSP_TEST_BGJOBS_WORKERS", "4")), thread_name_prefix="bgjob")
        _FUTURES[name] = _POOL.submit(_REGISTRY[name])


def result(name, timeout=3000):
    if not _enabled() and name not in _FUTURES:
        return _REGISTRY[name]()
    start(name)
    return _FUTURES[name].result(timeout=timeout)


def start_for(items):
    """tests/conftest.py, pytest_collection_finish: the jobs of the selected tests, in collection order"""
    if not _enabled():
        return
    for it in items:
        for name in _MODULE_JOBS.get(os.path.basename(str(getattr(it, "path", ""))), ()):
            start(name)
        names_of = getattr(getattr(it, "function", None), "_bgjobs", None)
        if names_of is None:
            continue
        params = dict(getattr(getattr(it, "callspec", None), "params", {}) or {})
        for name in names_of(params):
            start(name)


def shutdown():
    """tests/conftest.py, pytest_sessionfinish: jobs that have not started yet (a run that ended early: -x) are dropped"""
    global _POOL
    with _LOCK:
        if _POOL is not None:
            _POOL.shutdown(wait=False, cancel_futures=True)
            _POOL = None
