"""CPU: the host half of the library (parser, formatters, feature container, site enumerator, call_freq, gzip layer, fast5 reader) built
with AddressSanitizer + UBSan, and again with ThreadSanitizer (the parser / formatter thread teams, the parallel
inflater's decoder and finisher threads, the shared-memory ring), and driven over the fixtures plus mutated / truncated inputs
(tests/native/host_asan.cpp).  GPU sanitizers are not available on this pool; the HIP kernels are covered by the
parity suites instead."""
import os
import subprocess

from tests.helpers import GOLDEN, ROOT, cached_build

SRCS = ["dsp_text.cpp", "dsp_freq.cpp", "dsp_featfile.cpp", "dsp_sites.cpp", "dsp_gz.cpp", "dsp_fast5.cpp", "dsp_shmring.cpp", "dsp_pgz.cpp"]


import tempfile

import pytest

from tests import bgjobs

CSRC = os.path.join(ROOT, "deepsignal_plant_amd", "csrc")
SAN_ENV = dict(ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=0:exitcode=66")


def _san_cmd(sanitizers, driver, srcs):
    return ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=" + sanitizers] + (
        ["-fno-sanitize-recover=undefined"] if "undefined" in sanitizers else []) + ["-ffp-contract=off", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           os.path.join(ROOT, "tests", "native", driver)] + [os.path.join(CSRC, s) for s in srcs]


# the runs themselves are background jobs (tests/bgjobs.py): started when collection ends, next to the Python-level tests
for _san in ("address,undefined", "thread"):
    def _host(san=_san):
        exe = cached_build(_san_cmd(san, "host_asan.cpp", SRCS) + ["-lz", "-ldl", "-lrt"], "host_asan_" + san.replace(",", "_"))
        with tempfile.TemporaryDirectory(prefix="dsp_host_asan_") as d:
            return subprocess.run([exe, GOLDEN, d], capture_output=True, text=True, timeout=900, env=dict(os.environ, **SAN_ENV))

    def _refused(san=_san):
        exe = cached_build(_san_cmd(san, "threads_refused.cpp", ("dsp_text.cpp", "dsp_gz.cpp", "dsp_pgz.cpp")) + ["-lz", "-ldl", "-lrt"],
                           "threads_refused_" + san.replace(",", "_"))
        with tempfile.TemporaryDirectory(prefix="dsp_threads_refused_") as d:
            return subprocess.run([exe, d], capture_output=True, text=True, timeout=600, env=dict(os.environ, **SAN_ENV))
    bgjobs.job("host_asan_" + _san)(_host)
    bgjobs.job("threads_refused_" + _san)(_refused)


@bgjobs.job("parse_dev_host")
def _parse_dev_host():
    exe = cached_build(_san_cmd("address,undefined", "parse_dev_host.cpp", ("dsp_text.cpp",)), "parse_dev_host")
    return subprocess.run([exe, "200000", "30000"], capture_output=True, text=True, timeout=900, env=dict(os.environ, **SAN_ENV))


@pytest.mark.parametrize("sanitizers", ["address,undefined", "thread"])
@bgjobs.uses(lambda p: ["host_asan_" + p["sanitizers"]])
def test_host_code_is_clean_under_asan_and_ubsan(sanitizers):
    r = bgjobs.result("host_asan_" + sanitizers)
    assert r.returncode == 0 and "host_asan: ok" in r.stdout, (r.stdout[-2000:], r.stderr[-6000:])
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
    if os.path.exists("/opt/conda/lib/libhdf5.so"):  # the image's HDF5: the fast5 section must have run
        assert "fast5: " in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("sanitizers", ["address,undefined", "thread"])
@bgjobs.uses(lambda p: ["threads_refused_" + p["sanitizers"]])
def test_thread_pools_survive_a_system_that_refuses_threads(sanitizers):
    """csrc/dsp_threads.h: std::thread's constructor throws when the process may not have another thread (RLIMIT_NPROC, a pids
    cgroup), and an exception out of a pool of joinable threads -- or out of an extern "C" frame into ctypes -- is
    std::terminate: "Aborted", no message.  tests/native/threads_refused.cpp interposes pthread_create (EAGAIN after k
    calls) and drives the pools: every index of run_indexed still runs exactly once, BGZF deflate / inflate, the parallel
    inflater and the row parser give the bytes they give with all their threads, dsp_pgz_open without a decoder thread
    returns an error; a worker that throws is reported, not propagated.  Under ASan + UBSan, and under TSan."""
    r = bgjobs.result("threads_refused_" + sanitizers)
    assert r.returncode == 0 and "threads_refused: ok" in r.stdout, (r.stdout[-2000:], r.stderr[-6000:])
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
    assert int(r.stdout.split("(")[1].split()[0]) > 10      # the refusals really happened


@bgjobs.uses(lambda p: ["parse_dev_host"])
def test_device_parser_arithmetic_is_clean_under_asan_and_ubsan_and_equals_the_host_parser():
    """csrc/dsp_parse_arith.h is what dsp_parse_dev.hip compiles for gfx950 -- fast_float / fast_int / base_code / the SWAR
    delimiter masks and the per-token grammar (parse_token) -- and GPU sanitizers are not to be had on this pool.
    tests/native/parse_dev_host.cpp runs the token-parallel kernel's algorithm with that very source on the host under ASan +
    UBSan: 200,000 rows of random float spellings, 4,000 rows of the writer's grammar, 30,000 byte-mutated blocks.  Every row it
    accepts equals the host parser's bit for bit, every row the host parser rejects is flagged, plain rows are never flagged,
    and the sanitizers report nothing.  Test infrastructure: the product parses on the GPU only."""
    r = bgjobs.result("parse_dev_host")
    assert r.returncode == 0 and "parse_dev_host: ok" in r.stdout, (r.stdout[-2000:], r.stderr[-6000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-6000:]
    print(r.stdout)
    same = int(r.stdout.split("200000 rows, ")[1].split()[0])
    assert same > 100000      # (the rest: exponents beyond +-22, long mantissas, '+', blanks -- the host parser's)
    assert "never accepted what the host rejects" in r.stdout


def test_the_device_parser_compiles_the_shared_arithmetic_header():
    """... and the kernels really use it: no private copy of the number parsing left in the .hip"""
    src = open(os.path.join(ROOT, "deepsignal_plant_amd", "csrc", "dsp_parse_dev.hip")).read()
    assert '#include "dsp_parse_arith.h"' in src
    for name in ("bool fast_float(", "bool fast_int(", "int base_code(", "uint32_t eq_mask4(", "uint32_t delim_mask4(", "kPow10[23]"):
        assert name not in src, name
    assert "parse_token<LdsReader>(" in src
