"""DSP_SLOT_CANARY=1 -- a systematic check for the slot-reuse class of bug in the host pipeline.

The pipeline of `call_mods` passes two kinds of page-locked buffers round in rings: the reader's INPUT slots (reader thread
-> H2D / device parser -> forward -> writer thread -> back to the reader) and the RESULT slots (D2H of probabilities and
labels -> writer thread -> back to the main loop).  The bug found late in round 4 -- result slots going round by block
number, guarded by the event of the copy INTO the slot, which says nothing about the writer still formatting the slot's
previous block -- had passed 190 tests: nothing in the data says "this buffer was not yours to write".

In canary mode every slot is POISONED (all bytes 0xFF) the moment its owner releases it, the next owner VERIFIES that the
poison is intact when it takes the slot (a slot that is not all-0xFF was taken while its previous owner still held it, or
was written after it had been released), and every consumer verifies that what it is about to read holds NO poison
(0xFF never occurs in the ASCII text of a feature row; 0xFFFFFFFF is a NaN as float32, 255 is no base code and no label).
Cost: a memset and a scan per slot and hand-over -- a debugging mode for small inputs, off unless the variable is set.
"""
import os

import numpy as np


class SlotCanaryError(RuntimeError):
    pass


def on():
    return os.environ.get("DSP_SLOT_CANARY") == "1"


def _arrays(slot):
    if isinstance(slot, dict):
        for k, v in slot.items():
            if not str(k).startswith("_") and isinstance(v, np.ndarray):
                yield k, v
    else:
        for i, v in enumerate(slot):
            yield str(i), v


def _bytes(a):
    return np.ascontiguousarray(a).view(np.uint8).reshape(-1) if not a.flags["C_CONTIGUOUS"] else a.view(np.uint8).reshape(-1)


def poison(slot):
    """the owner lets go of `slot` (a dict of numpy arrays, or a list of them): every byte becomes 0xFF"""
    for _k, a in _arrays(slot):
        a.view(np.uint8).reshape(-1)[:] = 0xFF


def expect_poisoned(slot, what):
    """the next owner takes `slot`: nothing may have touched it since it was released"""
    for k, a in _arrays(slot):
        b = a.view(np.uint8).reshape(-1)
        if b.size and int(b.min()) != 0xFF:
            at = int(np.flatnonzero(b != 0xFF)[0])
            raise SlotCanaryError("DSP_SLOT_CANARY: %s: array %r is not poison at byte %d of %d -- the slot was taken while its "
                                  "previous owner still held it, or written after it was released" % (what, k, at, b.size))


def expect_live(named, what):
    """a consumer is about to read these (name, array) pairs: none of it may be poison -- it would be reading a slot that
    was already given back"""
    for k, a in named:
        if a is None or a.size == 0:
            continue
        b = a.view(np.uint8).reshape(-1) if a.flags["C_CONTIGUOUS"] else np.ascontiguousarray(a).view(np.uint8).reshape(-1)
        w = a.dtype.itemsize
        bad = (b.reshape(-1, w) == 0xFF).all(axis=1) if w > 1 else (b == 0xFF)
        if bool(bad.any()):
            raise SlotCanaryError("DSP_SLOT_CANARY: %s: %r holds poison at element %d of %d -- read after its slot was "
                                  "released" % (what, k, int(np.flatnonzero(bad)[0]), int(bad.size)))
