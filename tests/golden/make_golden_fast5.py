#!/opt/conda/bin/python3.9
"""Generate the F7 (fast5 ingestion) fixture by RUNNING THE REFERENCE on real HDF5 files (build container only).

Run with the image's second interpreter, the only one that has h5py and statsmodels:
    /opt/conda/bin/python3.9 tests/golden/make_golden_fast5.py
(h5py 3.3.0 over HDF5 1.10.6, statsmodels 0.12.2, numpy 1.26).  Nothing of the reference is replaced here: its
extract_features._extract_features opens the files with h5py, normalises with statsmodels' robust.mad, and its
_features_to_str prints the rows -- this is what pins the HDF5 decoding of this build (csrc/dsp_fast5.cpp) and the
MAD scale constant.

Writes under tests/golden/fast5/:
  reads/*.fast5         tombo-style single-read files written with h5py from deepsignal_plant_amd.reads.synth_reads
                        (seeded), in the storage variants met in the wild: chunked + gzip + shuffle / contiguous
                        Signal, fixed-length byte and variable-length UTF-8 string attributes, u4 and i8 event columns,
                        integer-typed channel offset; plus files the reference counts as errors (no Alignment group,
                        no read_start_rel_to_raw, no Raw/Reads, not an HDF5 file)
  expect_<case>.tsv.gz  the rows the reference's _extract_features + _features_to_str produce, in sorted-file order
  expect.json           per case: arguments, row count, error count; per file: what h5py returned (checksums)
"""
import gzip
import importlib.util
import json
import os
import random
import shutil
import sys

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "fast5")
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from deepsignal_plant import extract_features as ref  # noqa: E402  (the reference, unmodified)
from deepsignal_plant.utils.process_utils import get_motif_seqs, parse_region_str  # noqa: E402

spec = importlib.util.spec_from_file_location("dsp_reads", os.path.join(ROOT, "deepsignal_plant_amd", "reads.py"))
dsp_reads = importlib.util.module_from_spec(spec)
spec.loader.exec_module(dsp_reads)

CHROM_LEN = {"chr1": 40000, "chr2": 36000, "chr3": 30011}
EVENT_DT = [("norm_mean", "<f8"), ("norm_stdev", "<f8"), ("start", "<u4"), ("length", "<u4"), ("base", "S1")]
EVENT_DT_WIDE = [("norm_mean", "<f8"), ("norm_stdev", "<f8"), ("start", "<i8"), ("length", "<i8"), ("base", "S1")]


def write_fast5(path, r, variant):
    """variant: 0 fixed-length byte attributes, chunked gzip+shuffle signal (what tombo + ont_fast5_api write);
    1 variable-length UTF-8 attributes, contiguous signal, i8 event columns; 2 as 0 with an integer channel offset and
    an extra read-independent group in Raw/Reads' parent; 3.. broken on purpose (see below)"""
    rel = int(r.ev_start[0])
    ev = np.zeros(len(r.ev_base), dtype=EVENT_DT_WIDE if variant == 1 else EVENT_DT)
    ev["start"] = r.ev_start - rel
    ev["length"] = r.ev_len
    ev["base"] = r.ev_base.view("S1")
    ev["norm_mean"] = 0.0
    ev["norm_stdev"] = 1.0
    with h5py.File(path, "w") as f:
        f.attrs["file_version"] = 2.0
        s = (lambda x: x) if variant == 1 else np.string_
        if variant != 5:
            rd = f.create_group("Raw/Reads/Read_%d" % (17 + variant))
            rd.attrs["read_id"] = s(r.readname)
            rd.attrs["read_number"] = np.int32(17 + variant)
            rd.attrs["start_time"] = np.uint64(123456)
            rd.attrs["duration"] = np.uint32(len(r.raw))
            if variant == 1:
                rd.create_dataset("Signal", data=r.raw)
            else:
                rd.create_dataset("Signal", data=r.raw, chunks=(min(len(r.raw), 1024),), compression="gzip",
                                  compression_opts=1, shuffle=(variant != 2))
        ch = f.create_group("UniqueGlobalKey/channel_id")
        ch.attrs["digitisation"] = np.float64(8192.0)
        ch.attrs["range"] = np.float64(r.scaling * 8192.0)
        ch.attrs["offset"] = np.int64(r.offset) if variant == 2 else np.float64(r.offset)
        ch.attrs["sampling_rate"] = np.float64(4000.0)
        ch.attrs["channel_number"] = s("101")
        f.create_group("UniqueGlobalKey/tracking_id").attrs["run_id"] = s("synthetic")
        sub = f.create_group("Analyses/RawGenomeCorrected_000/BaseCalled_template")
        f["Analyses/RawGenomeCorrected_000"].attrs["tombo_version"] = s("1.5.1")
        sub.attrs["status"] = s("success")
        d = sub.create_dataset("Events", data=ev, compression="gzip" if variant != 1 else None)
        if variant != 4:
            d.attrs["read_start_rel_to_raw"] = np.int64(rel)
        if variant != 3:
            al = sub.create_group("Alignment")
            al.attrs["mapped_chrom"] = s(r.chrom)
            al.attrs["mapped_strand"] = s(r.alignstrand)
            al.attrs["mapped_start"] = np.int64(r.chrom_start)
            al.attrs["mapped_end"] = np.int64(r.chrom_start + len(r.ev_base))
            al.attrs["num_matches"] = np.int64(len(r.ev_base))
        # a second corrected group that must be ignored
        f.create_group("Analyses/Basecall_1D_000/BaseCalled_template").attrs["status"] = s("ok")


def reads_for_fixture():
    rs = dsp_reads.synth_reads(14, seed=2024, mean_bases=330)
    rng = np.random.default_rng(7)
    for r in rs:
        r.chrom_start = int(rng.integers(0, CHROM_LEN[r.chrom] - 2 * len(r.ev_base) - 10))
    return rs


CASES = [
    dict(name="mad_cg", method="mad", motifs="CG", mod_loc=0, k=13, s=16, label=1, c2l=True, region=None),
    dict(name="zscore_chg_k9", method="zscore", motifs="CHG", mod_loc=0, k=9, s=12, label=0, c2l=False, region=None),
    dict(name="mad_region", method="mad", motifs="CG", mod_loc=0, k=13, s=16, label=1, c2l=True, region="chr2:1000-30000"),
]


def main():
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(os.path.join(OUT, "reads", "sub"))
    rs = reads_for_fixture()
    files = []
    for i, r in enumerate(rs):
        variant = {9: 3, 10: 4, 11: 5}.get(i, i % 3)
        p = os.path.join(OUT, "reads", "sub" if i % 4 == 3 else "", "%s_ch101_read%d_strand.fast5" % (r.readname, i))
        write_fast5(p, r, variant)
        files.append((p, variant))
    with open(os.path.join(OUT, "reads", "not_hdf5.fast5"), "wb") as f:
        f.write(b"this is not an HDF5 file\n" * 40)
    files.append((os.path.join(OUT, "reads", "not_hdf5.fast5"), 6))
    fast5s = sorted(p for p, _ in files)
    ref_fa = os.path.join(OUT, "_ref.fa")  # not kept: the tests rebuild it from expect.json's chrom_len
    with open(ref_fa, "w") as f:
        for c, n in CHROM_LEN.items():
            f.write(">%s some description\n" % c)
            for o in range(0, n, 60):
                f.write("N" * min(60, n - o) + "\n")
    expect = {"chrom_len": CHROM_LEN, "files": {}, "cases": {},
              "versions": {"h5py": h5py.__version__, "hdf5": h5py.version.hdf5_version, "numpy": np.__version__,
                           "statsmodels": __import__("statsmodels").__version__}}
    for p, variant in files:
        rel = os.path.relpath(p, os.path.join(OUT, "reads"))
        info = {"variant": variant}
        try:
            raw, events = ref._get_label_raw(p, "RawGenomeCorrected_000", "BaseCalled_template")
            al = ref._get_alignment_info_from_fast5(p, "RawGenomeCorrected_000", "BaseCalled_template")
            sc = ref._get_scaling_of_a_read(p)
            info.update(ok=True, n_raw=int(len(raw)), raw_sum=int(np.asarray(raw, np.int64).sum()),
                        n_events=len(events), start_sum=int(sum(int(e[0]) for e in events)),
                        len_sum=int(sum(int(e[1]) for e in events)), seq="".join(e[2] for e in events),
                        readname=al[0], strand=al[1], alignstrand=al[2], chrom=al[3],
                        chrom_start=(int(al[4]) if al[4] != "" else None), scaling=float(sc[0]), offset=float(sc[1]))
        except Exception as e:  # the reference counts these files as errors
            info.update(ok=False, error=type(e).__name__)
        expect["files"][rel] = info
    for c in CASES:
        motif_seqs = get_motif_seqs(c["motifs"], True)
        chrom2len = ref.get_contig2len(ref_fa) if c["c2l"] else None
        random.seed(1234)
        feats, nerr = ref._extract_features(fast5s, "RawGenomeCorrected_000", "BaseCalled_template", c["method"], motif_seqs,
                                            c["mod_loc"], chrom2len, c["k"], c["s"], c["label"], None,
                                            parse_region_str(c["region"]))
        rows = [ref._features_to_str(f) for f in feats]
        with open(os.path.join(OUT, "expect_%s.tsv.gz" % c["name"]), "wb") as raw:
            with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, filename="") as f:
                f.write(("\n".join(rows) + "\n").encode())
        expect["cases"][c["name"]] = dict(c, rows=len(rows), errors=int(nerr))
        print("%-16s %5d rows, %d of %d files failed" % (c["name"], len(rows), nerr, len(fast5s)))
    # ---- the reference's real command line (`deepsignal_plant extract`, multi-process) on the same directory
    import glob
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp()
    first = [r for r in gzip.open(os.path.join(OUT, "expect_mad_cg.tsv.gz"), "rt").read().splitlines()]
    pos_file = os.path.join(OUT, "positions.tsv")
    with open(pos_file, "w") as f:  # every third site of the plain run (chrom, pos, strand + a column that is ignored)
        for row in sorted(set("\t".join(r.split("\t")[:3]) for r in first))[::3]:
            f.write(row + "\tx\n")
    cli = {
        "cli_plain": ["--reference_path", ref_fa, "--nproc", "3", "--f5_batch_size", "4"],
        "cli_dir_gzip": ["--reference_path", ref_fa, "--nproc", "2", "--f5_batch_size", "3", "--w_is_dir", "yes", "--w_batch_num", "2", "--gzip"],
        "cli_positions": ["--reference_path", ref_fa, "--nproc", "2", "--positions", pos_file, "--methy_label", "0", "--normalize_method", "zscore"],
    }
    expect["cli"] = {}
    for name, flags in cli.items():
        out = os.path.join(tmp, name + (".d" if "--w_is_dir" in flags else ".tsv"))
        env = dict(os.environ, PYTHONPATH="/root/reference", PYTHONDONTWRITEBYTECODE="1")
        r = subprocess.run([sys.executable, "-m", "deepsignal_plant.deepsignal_plant", "extract", "-i", os.path.join(OUT, "reads"), "-o", out] + flags,
                           capture_output=True, text=True, env=env, cwd=tmp)
        assert r.returncode == 0, r.stderr[-2000:]
        rows = []
        if os.path.isdir(out):
            names = sorted(os.listdir(out))
            for fn in names:
                rows += gzip.open(os.path.join(out, fn), "rt").read().splitlines()
        else:
            names = [os.path.basename(out)]
            rows = open(out).read().splitlines()
        with open(os.path.join(OUT, "expect_%s.tsv.gz" % name), "wb") as raw:
            with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, filename="") as f:
                f.write(("\n".join(sorted(rows)) + "\n").encode())
        expect["cli"][name] = {"flags": [("<ref.fa>" if x == ref_fa else "<positions.tsv>" if x == pos_file else x) for x in flags],
                               "rows": len(rows), "files": names}
        print("%-16s %5d rows in %d file(s)" % (name, len(rows), len(names)))
    shutil.rmtree(tmp, ignore_errors=True)
    os.remove(ref_fa)
    with open(os.path.join(OUT, "expect.json"), "w") as f:
        json.dump(expect, f, indent=1, sort_keys=True)
    os.system("du -sh %s" % OUT)


if __name__ == "__main__":
    main()
