#!/usr/bin/env python3
"""Build-time gate for a gfx950 store-data hazard that hipcc (ROCm 7.2) does not pad.

Measured on MI355X (tools/micro/h_exchange.hip, profiles/r3/micro_h_exchange.txt): a VMEM store of more than 64 bits
followed IMMEDIATELY by a VALU write of its data registers stores the NEW values for the last quad of every 16-lane row
(lanes 12-15, 28-31, 44-47, 60-63), depending on what the SIMD's other wave is doing:

    buffer_store_dwordx4 v[4:7], v17, s[8:11], s16 offen      ; SGPR soffset
    v_pk_add_f32 v[6:7], ...                                   ; 0 wait states later: CORRUPTS the store  (1 is enough)

    buffer_store_dwordx4 v[4:7], v17, s[8:11], 0 offen        ; immediate soffset: the case the ISA manual documents
    s_nop 0 / v_mov_b32 v5, ...                                ; 0 or 1 wait states later: corrupts       (2 are enough)

LLVM's hazard recognizer pads the second form (GCNHazardRecognizer::createsVALUHazard, 2 wait states on gfx940+) but
assumes that an SGPR soffset hides the hazard, which on gfx950 it does not: hipcc emits the first form unpadded (it did so
in the micro-benchmark's own check kernel).  This is the round-2 "h0 read-back" fault of the <2 unit tiles, 1 site tile>
LSTM tiling: same lanes, non-deterministic, second wave of a SIMD, present or absent depending on code generation.
MFMA, transcendental and VMEM-load writers did not corrupt the store in the measurement; they are checked all the same.

This script scans device assembly (hipcc -S --cuda-device-only) and fails if any wide VMEM store is followed by a VALU
write of one of its data VGPRs within fewer than TWO wait states: with an SGPR soffset that is the measured need (1) plus
one of margin; in the other forms (immediate soffset, global / scratch stores -- e.g. the compiler's own register spills)
it is the measured need and exactly what the compiler's hazard recognizer pads.  Writers that return data from memory
(VMEM / LDS loads into the registers) never corrupted a store in the measurement -- the data has long left the register
file when they land; they are listed with --verbose only.
The csrc Makefile runs it on every .hip translation unit of the library; tests/test_store_hazard.py runs it on the CPU.

usage: check_store_hazard.py [--verbose] FILE.s [FILE.s ...]      (exit code 1 and a listing when a hazard is found)
"""
import re
import sys

WIDE_STORE = re.compile(r"^\s*(buffer_store_dwordx[34]|global_store_dwordx[34]|flat_store_dwordx[34]|scratch_store_dwordx[34]|"
                        r"buffer_store_format_xyzw?|tbuffer_store_format_xyzw?)\b(.*)$")
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
INSTR = re.compile(r"^\s*([a-z_][a-z0-9_]*)\s*(.*)$")


def regs_of(token):
    m = VREG.search(token)
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def first_operand(ops):
    return ops.split(",")[0].strip() if ops.strip() else ""


def wait_states(mn, ops):
    """wait states an instruction contributes between the store and a later writer"""
    if mn == "s_nop":
        try:
            return int(ops.split()[0], 0) + 1
        except (ValueError, IndexError):
            return 1
    return 1


def scan(path):
    findings = []
    func = "?"
    lines = open(path, errors="replace").read().splitlines()
    code = []   # (line number, function, mnemonic, operands, raw)
    for no, raw in enumerate(lines, 1):
        line = raw.split(";")[0].rstrip()
        if not line.strip():
            continue
        if re.match(r"^[A-Za-z_.$][\w.$]*:\s*$", line.strip()):
            if not line.strip().startswith(".L"):
                func = line.strip()[:-1]
            continue
        if line.lstrip().startswith("."):
            continue
        m = INSTR.match(line)
        if m:
            code.append((no, func, m.group(1), m.group(2), raw.strip()))
    for i, (no, fn, mn, ops, raw) in enumerate(code):
        ms = WIDE_STORE.match(mn + " " + ops)
        if not ms:
            continue
        parts = [p.strip() for p in ops.split(",")]
        if mn.startswith("global_store") or mn.startswith("flat_store") or mn.startswith("scratch_store"):
            data = regs_of(parts[1]) if len(parts) > 1 else set()   # global_store vaddr, vdata, saddr
        else:
            data = regs_of(parts[0])
        need = 2
        if len(data) < 3:
            continue
        waited = 0
        for (no2, fn2, mn2, ops2, raw2) in code[i + 1:i + 8]:
            if fn2 != fn or waited >= need:
                break
            writes, valu = set(), False
            if mn2.startswith("v_") and not mn2.startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane")):
                writes = regs_of(first_operand(ops2)) if first_operand(ops2).startswith("v") else set()
                valu = True
            elif mn2.startswith(("buffer_load", "global_load", "flat_load", "scratch_load", "ds_read", "ds_load")):
                writes = regs_of(first_operand(ops2)) if first_operand(ops2).startswith("v") else set()
            if writes & data:
                findings.append((path, fn, no, raw, no2, raw2, waited, need, valu))
                break
            waited += wait_states(mn2, ops2)
    return findings, sum(1 for c in code if WIDE_STORE.match(c[2] + " " + c[3]))


def main(argv):
    verbose = "--verbose" in argv
    argv = [a for a in argv if a != "--verbose"]
    if not argv:
        raise SystemExit(__doc__)
    found, stores = [], 0
    for p in argv:
        f, n = scan(p)
        found += f
        stores += n
    bad = [f for f in found if f[8]]
    for path, fn, no, raw, no2, raw2, waited, need, valu in (found if verbose else bad):
        print("%s:%d: in %s\n    %s\n    %s   <- line %d %s the store's data registers after %d wait state(s)%s"
              % (path, no, fn, raw, raw2, no2, "writes" if valu else "loads into", waited,
                 "; %d needed" % need if valu else " (a memory return: benign)"))
    print("check_store_hazard: %d wide VMEM stores scanned in %d file(s), %d hazard(s), %d early loads into store data registers"
          % (stores, len(argv), len(bad), len(found) - len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
