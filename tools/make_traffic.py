#!/usr/bin/env python3
"""One entry of profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE PMC summaries of tools/profile.sh: HBM-side
bytes per launch of the dominant kernel (the non-SPARSE dsp_lstm_kernel instantiation = the combined stack), stamped
with the hash of the kernel sources and the workload, so that bench.py only quotes it while both still match.
usage: make_traffic.py PROF_DIR [bench.py flags]"""
import json
import os
import re
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def avg_of(path, kernel_re, counter):
    for line in open(path):
        if re.search(kernel_re, line) and counter in line:
            return float(line.split("avg=")[1].split()[0])
    raise SystemExit("no %s line for %s in %s" % (counter, kernel_re, path))


def all_kernels_per_forward(path, counter):
    """sum of `counter` over the library's own kernels (dsp_*), per forward (= per dsp_head_kernel dispatch)"""
    total, forwards = 0.0, 0
    for line in open(path):
        if counter not in line or "dsp_" not in line.split("dispatches=")[0]:
            continue
        total += float(line.split("sum=")[1].split()[0])
        if "dsp_head_kernel" in line:
            forwards = int(line.split("dispatches=")[1].split()[0])
    if not forwards:
        raise SystemExit("no dsp_head_kernel line in %s" % path)
    return total / forwards


def main():
    d = sys.argv[1]
    args = bench.parse_args(sys.argv[2:])
    kern = r"dsp_lstm_kernel<(false|0),"
    fetch_kb = avg_of(os.path.join(d, "pmc_FETCH_SIZE.txt"), kern, "FETCH_SIZE")
    write_kb = avg_of(os.path.join(d, "pmc_WRITE_SIZE.txt"), kern, "WRITE_SIZE")
    entry = {"model_type": args.model_type, "layernum1": args.layernum1, "hid_rnn": args.hid_rnn, "batch": args.batch,
             "precision": args.precision, "kernel_src_sha16": bench.kernel_source_hash(),
             "kernel": "dsp_lstm_kernel<0, 1> (combined stack)",
             "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 2 --warmup 1 "
                       "--no_cpu_baseline --no_alt %s`, tools/profile.sh" % " ".join(sys.argv[2:]),
             "fetch_size_kb_per_launch": fetch_kb, "write_size_kb_per_launch": write_kb,
             "correction": "FETCH_SIZE doubled (gfx950 counts the 128-B requests of wide coalesced reads as 64 B, "
                           "MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported (uncalibrated); Infinity-Cache hits are included",
             "hbm_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
             # all nine launches of one forward (pack, front ends, fc, combined stack, head), same correction
             "hbm_bytes_per_step_all_kernels": 2 * 1024 * all_kernels_per_forward(os.path.join(d, "pmc_FETCH_SIZE.txt"), "FETCH_SIZE")
             + 1024 * all_kernels_per_forward(os.path.join(d, "pmc_WRITE_SIZE.txt"), "WRITE_SIZE")}
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
