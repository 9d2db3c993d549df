#!/usr/bin/env python3
"""Compare two builds' gfx950 device assembly kernel by kernel -- how round 6 checked kernel changes with no GPU to run them on.

    make -C deepsignal_plant_amd/csrc            # leaves _obj/<unit>-hip-amdgcn-amd-amdhsa-gfx950.s
    python tools/asm_equivalence.py OLD.s NEW.s

Per kernel: instruction counts, the opcodes whose counts differ (scalar bookkeeping apart: anything that is not s_* / v_cmp /
v_mov / v_cndmask / v_readfirstlane, and every s_waitcnt / s_nop, is listed), SGPR spills (v_readlane / v_writelane), and
whether the OPCODE STREAM FROM THE FIRST MFMA ON (the step loop and everything behind it; s_waitcnt with its vmcnt count) is
identical.  Register numbers are not compared: the allocator may rename without changing a cycle.
To get OLD.s of an earlier commit: git worktree add /tmp/old <commit>; make -C /tmp/old/deepsignal_plant_amd/csrc _obj/dsp_kernels.o"""
import collections
import difflib
import re
import sys


def parse(path):
    fn, out = None, collections.OrderedDict()
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            fn = m.group(1)
            out[fn] = []
            continue
        if fn and line.startswith("\t") and not line.startswith("\t.") and not line.startswith("\t;"):
            ins = line.strip().split(";")[0].strip()
            if ins:
                out[fn].append(ins)
    return {k: v for k, v in out.items() if v}


def stream(ins):
    ops = [i.split()[0] + (" vmcnt" + i.split("vmcnt")[1][:4] if "vmcnt" in i else "") for i in ins]
    first = next((k for k, o in enumerate(ops) if o.startswith("v_mfma")), None)
    return ops if first is None else ops[first:]


def main():
    a, b = parse(sys.argv[1]), parse(sys.argv[2])
    scalar = re.compile(r"^(s_|v_cmp|v_mov_b|v_cndmask|v_readfirstlane)")
    for fn in a:
        if fn not in b:
            print("%-64s only in %s" % (fn[:64], sys.argv[1]))
            continue
        ca, cb = (collections.Counter(i.split()[0] for i in x[fn]) for x in (a, b))
        d = {k: cb[k] - ca[k] for k in set(ca) | set(cb) if cb[k] != ca[k]}
        odd = {k: v for k, v in sorted(d.items()) if not scalar.match(k) or k in ("s_waitcnt", "s_nop")}
        x, y = stream(a[fn]), stream(b[fn])
        if x == y:
            tail = "from the first MFMA on: IDENTICAL (%d instructions)" % len(x)
        else:
            n = sum(1 for t in difflib.SequenceMatcher(None, x, y, autojunk=False).get_opcodes() if t[0] != "equal")
            tail = "from the first MFMA on: differs in %d places (%d -> %d)" % (n, len(x), len(y))
        print("%-64s %5d -> %5d  other than scalar: %s  SGPR spill reads %d -> %d  %s" % (
            fn[:64], len(a[fn]), len(b[fn]), odd or "-", ca["v_readlane_b32"], cb["v_readlane_b32"], tail))
    for fn in b:
        if fn not in a:
            print("%-64s only in %s" % (fn[:64], sys.argv[2]))


if __name__ == "__main__":
    main()
