"""CPU: the off-GPU model of the clustered LSTM launches' hand-off protocol (tests/native/cluster_model.cpp).

The admission of a cluster's members and the wait for a step's arrivals are compiled from the kernel's own source
(csrc/dsp_cluster_protocol.h, an Ops policy: device atomics in dsp_kernels.hip, std::atomic here); the per-step skeleton around
them -- drain, barrier, arrival per member (round 4) or per wave and deferred into the next step's x part (round 5), the LDS gate
exchange, the clean-up launch -- is transcribed with the kernel's line references.  Two engines: real threads under
ThreadSanitizer (h rows and the LDS exchange are plain memory: what the protocol does not order is a data race), and a
single-threaded explorer with a store buffer per wave and adversarial residency, 1e6 launches over P = 2 / 4 / 8 members: a
member resident late or only after the others gave up, a wave held up mid-step, two forwards on the same counters.
Every mutant must fail, or the model checks nothing."""
import os
import subprocess
import sys

import pytest

from tests import bgjobs
from tests.helpers import ROOT, cached_build

SRC = os.path.join(ROOT, "tests", "native", "cluster_model.cpp")
INC = os.path.join(ROOT, "deepsignal_plant_amd", "csrc")
MUTANTS = ["nodrain", "latearrive", "nozero", "twice", "never"]



BASE = ["g++", "-std=c++17", "-Wall", "-Werror", "-pthread", "-I", INC, SRC]


def _fast():
    return cached_build(BASE + ["-O2"], "cluster_model")


def _tsan():
    return cached_build(BASE + ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=thread"], "cluster_model_tsan")


# every run is a background job (tests/bgjobs.py): started when collection ends, next to the Python-level tests
@bgjobs.job("cluster_explore")
def _explore():
    fast = _fast()
    procs = [subprocess.Popen([fast, "explore", "62500", "none", str(k * 10_000_000)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for k in range(4)]
    return [(p.communicate(timeout=600)[0], p.returncode) for p in procs]


@bgjobs.job("cluster_tsan_threads")
def _tsan_threads():
    return subprocess.run([_tsan(), "threads", "120"], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0:exitcode=66"))


for _m in MUTANTS:
    bgjobs.job("cluster_explore_" + _m)(lambda m=_m: subprocess.run([_fast(), "explore", "20000", m], capture_output=True, text=True, timeout=300))
    bgjobs.job("cluster_tsan_" + _m)(lambda m=_m: subprocess.run([_tsan(), "threads", "150", m], capture_output=True, text=True, timeout=600,
                                                                  env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66")))


@bgjobs.uses(lambda p: ["cluster_explore"])
def test_a_million_scheduled_launches_hold_every_invariant():
    """No deadlock within the tick budget; in a cluster that is not abandoned every wave runs to its last step, every row read
    behind a passed poll is complete, no LDS gate slot is read stale, the rows are the sequential reference's; an abandoned
    cluster is recomputed exactly once by the clean-up launch, a cluster that ran is never recomputed; without adversity no
    cluster is abandoned."""
    launches = abandoned = 0
    for out, rc in bgjobs.result("cluster_explore"):
        assert rc == 0 and "cluster_model: ok" in out, out[-2000:]
        launches += int(out.split("= ")[1].split()[0])
        abandoned += int(out.split("clean, ")[1].split()[0])
    print("explorer: %d launches scheduled, %d abandoned and recomputed" % (launches, abandoned))
    assert launches >= 1_000_000 and 0.05 * launches < abandoned < 0.6 * launches   # (the give-up paths are really exercised)


@pytest.mark.parametrize("mutant,expect", [
    ("nodrain", "before every part of it was in memory"),          # the arrival counted with the h stores still in flight
    ("latearrive", "no progress within the tick budget"),          # every wave polls for an arrival it has not made yet
    ("nozero", "before every part of it was in memory"),           # stale counters admit at once, every arrival "already in"
    ("twice", "recomputed a cluster that was not abandoned"),
    ("never", "wrong h row behind the launch"),
])
@bgjobs.uses(lambda p: ["cluster_explore_" + p["mutant"]])
def test_the_explorer_fails_on_every_mutant(mutant, expect):
    r = bgjobs.result("cluster_explore_" + mutant)
    assert r.returncode == 1 and "VIOLATION" in r.stdout and expect in r.stdout, (r.returncode, r.stdout[-1500:])


@bgjobs.uses(lambda p: ["cluster_tsan_threads"])
def test_threads_under_tsan_report_nothing():
    r = bgjobs.result("cluster_tsan_threads")
    assert r.returncode == 0 and "cluster_model: ok" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-4000:])
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
    print(r.stdout.strip())


@pytest.mark.parametrize("mutant", MUTANTS)
@bgjobs.uses(lambda p: ["cluster_tsan_" + p["mutant"]])
def test_the_threads_engine_fails_on_every_mutant(mutant):
    """nodrain = the arrival's add without release semantics (the device: no s_waitcnt vmcnt(0) in front of it): a data race on the
    h rows that only ThreadSanitizer can see on an x86 host -- and does."""
    r = bgjobs.result("cluster_tsan_" + mutant)
    race = "WARNING: ThreadSanitizer: data race" in r.stderr
    assert r.returncode != 0 and (race or "VIOLATION" in r.stdout), (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    if mutant == "nodrain":
        assert race, r.stderr[-3000:]
