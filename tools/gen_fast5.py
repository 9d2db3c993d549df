#!/opt/conda/bin/python3.9
"""Write synthetic tombo-style single-read fast5 files with h5py (benchmark input; run with the image's python3.9, the
interpreter that has h5py): gen_fast5.py OUT_DIR N_READS MEAN_BASES [CG_BOOST].  Same writer as the F7 fixture
(tests/golden/make_golden_fast5.py: chunked gzip + shuffle signal, gzip events table)."""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    out, n, mean_bases = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    cg = float(sys.argv[4]) if len(sys.argv) > 4 else 0.15
    import h5py  # noqa: F401  (fail early with a clear message if this is the wrong interpreter)
    import numpy as np
    R = load("dsp_reads", os.path.join(ROOT, "deepsignal_plant_amd", "reads.py"))
    ev_dt = [("norm_mean", "<f8"), ("norm_stdev", "<f8"), ("start", "<u4"), ("length", "<u4"), ("base", "S1")]
    os.makedirs(out, exist_ok=True)
    samples = bases = 0
    for i in range(0, n, 64):
        for j, r in enumerate(R.synth_reads(min(64, n - i), seed=100 + i, mean_bases=mean_bases, cg_boost=cg)):
            d = os.path.join(out, "%03d" % (i // 64))
            os.makedirs(d, exist_ok=True)
            rel = int(r.ev_start[0])
            ev = np.zeros(len(r.ev_base), dtype=ev_dt)
            ev["start"], ev["length"], ev["base"] = r.ev_start - rel, r.ev_len, r.ev_base.view("S1")
            with h5py.File(os.path.join(d, "%s.fast5" % r.readname), "w") as f:
                rd = f.create_group("Raw/Reads/Read_%d" % (i + j))
                rd.attrs["read_id"] = np.string_(r.readname)
                rd.create_dataset("Signal", data=r.raw, chunks=(min(len(r.raw), 4096),), compression="gzip", compression_opts=1, shuffle=True)
                ch = f.create_group("UniqueGlobalKey/channel_id")
                ch.attrs["digitisation"], ch.attrs["range"], ch.attrs["offset"] = np.float64(8192.0), np.float64(r.scaling * 8192.0), np.float64(r.offset)
                sub = f.create_group("Analyses/RawGenomeCorrected_000/BaseCalled_template")
                e = sub.create_dataset("Events", data=ev, compression="gzip")
                e.attrs["read_start_rel_to_raw"] = np.int64(rel)
                al = sub.create_group("Alignment")
                al.attrs["mapped_chrom"], al.attrs["mapped_strand"] = np.string_(r.chrom), np.string_(r.alignstrand)
                al.attrs["mapped_start"] = np.int64(r.chrom_start)
            samples += len(r.raw)
            bases += len(r.ev_base)
    print("%d %d" % (samples, bases))


if __name__ == "__main__":
    main()
