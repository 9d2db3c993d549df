"""CPU: the slot canary (deepsignal_plant_amd/canary.py, DSP_SLOT_CANARY=1) on the reader's slot ring -- the host half of
tests/test_gpu_canary.py.  The GPU is not needed to show what it catches: a consumer that keeps reading a block it has
released, and a consumer that writes into a slot it no longer owns."""
import os

import numpy as np
import pytest

from deepsignal_plant_amd import canary, feed
from tests.helpers import GOLDEN

ROWS = os.path.join(GOLDEN, "f2_rows.tsv")


def test_poison_and_the_two_checks():
    slot = {"a": np.zeros((4, 3), np.float32), "b": np.zeros(5, np.uint8), "_private": np.zeros(3), "cap_rows": 4}
    canary.poison(slot)
    assert np.isnan(slot["a"]).all() and (slot["b"] == 255).all() and (slot["_private"] == 0).all()
    canary.expect_poisoned(slot, "take")
    slot["b"][4] = 7
    with pytest.raises(canary.SlotCanaryError, match="array 'b' is not poison at byte 4 of 5"):
        canary.expect_poisoned(slot, "take")
    canary.poison(slot)
    slot["a"][:2] = 1.5
    canary.expect_live([("a", slot["a"][:2]), ("nothing", None), ("empty", slot["a"][:0])], "use")
    with pytest.raises(canary.SlotCanaryError, match="'a' holds poison at element 6 of 9"):
        canary.expect_live([("a", slot["a"][:3])], "use")
    # a float32 NaN that a row can really hold (float("nan")) is not poison; a lens value of -1 (0xFFFFFFFF) would be
    canary.expect_live([("nan", np.array([np.nan], np.float32))], "use")
    assert not canary.on() or os.environ.get("DSP_SLOT_CANARY") == "1"


def _reader(monkeypatch, canary_on=True):
    if canary_on:
        monkeypatch.setenv("DSP_SLOT_CANARY", "1")
    return feed.FeatureReader(ROWS, 13, 16, nthreads=2, nbuf=3, block_bytes=40_000, pinned=False)


def test_a_well_behaved_consumer_passes_and_sees_the_rows(monkeypatch):
    r = _reader(monkeypatch)
    assert r.canary
    r.start()
    n, sums = 0, 0.0
    for blk in r:
        canary.expect_live([("means", blk.rows.means), ("kmer", blk.rows.kmer), ("signals", blk.rows.signals)], "consumer")
        n += blk.rows.n
        sums += float(blk.rows.means.sum())
        r.release(blk)
    assert n == 200 and np.isfinite(sums)
    plain = feed.FeatureReader(ROWS, 13, 16, nthreads=2, nbuf=3, block_bytes=40_000, pinned=False)
    assert not plain.canary or True
    plain.canary = False
    plain.start()
    total = 0.0
    for b in plain:
        total += float(b.rows.means.sum())
        plain.release(b)
    assert abs(total - sums) < 1e-3


def test_reading_a_block_after_releasing_it_is_caught(monkeypatch):
    r = _reader(monkeypatch)
    r.start()
    it = iter(r)
    blk = next(it)
    means = blk.rows.means              # a view of the slot
    r.release(blk)
    with pytest.raises(canary.SlotCanaryError, match="read after its slot was released"):
        canary.expect_live([("means", means)], "a consumer that kept a view of a released block")
    for b in it:
        r.release(b)


def test_writing_into_a_released_slot_is_caught_by_the_next_owner(monkeypatch):
    r = _reader(monkeypatch)
    r.start()
    it = iter(r)
    blocks = [next(it), next(it)]
    stale = blocks[0].slot
    r.release(blocks[0])
    stale["labels"][0] = 3              # e.g. a late asynchronous copy landing in a slot that was handed back
    r.release(blocks[1])
    with pytest.raises(canary.SlotCanaryError, match="reader takes an input slot: array 'labels' is not poison"):
        for b in it:                    # the reader thread's error surfaces in the consumer
            r.release(b)
