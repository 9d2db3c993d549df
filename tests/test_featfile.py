"""CPU: the binary feature container (.dspf) holds exactly what the reference's reader yields for the same
rows (fixture F2, captured from _read_features_file by tests/golden/make_golden_text.py), block by block."""
import os
import subprocess
import sys

import numpy as np
import pytest

from deepsignal_plant_amd import featfile, feed, textio
from tests.helpers import GOLDEN, ROOT

KEYS = ("kmer", "means", "stds", "lens", "signals", "labels")


def _all_rows(path):
    with featfile.FeatureFile(path) as ff:
        blocks = [ff.read_block(b, nthreads=3)[0] for b in range(ff.n_blocks)]
        meta = (ff.seq_len, ff.signal_len, ff.n_rows, ff.n_blocks, ff.block_n.copy(), ff.block_first_row.copy())
    cat = {k: np.concatenate([getattr(r, k) for r in blocks]) for k in KEYS}
    info = [r.sampleinfo(i) for r in blocks for i in range(r.n)]
    reads = [r.readname(i) for r in blocks for i in range(r.n)]
    return cat, info, reads, meta


@pytest.mark.parametrize("gz", [False, True])
@pytest.mark.parametrize("block_rows", [7, 64, 32768])
def test_pack_roundtrip_matches_reference_reader(tmp_path, gz, block_rows):
    f2 = np.load(os.path.join(GOLDEN, "f2_parsed.npz"))
    src = os.path.join(GOLDEN, "f2_rows.tsv" + (".gz" if gz else ""))
    dst = os.path.join(str(tmp_path), "f2.dspf")
    n = featfile.pack_features(src, dst, 13, 16, block_rows=block_rows, nthreads=2, chunk_bytes=100000)
    assert n == 200 and featfile.is_feature_file(dst) and not featfile.is_feature_file(src)
    cat, info, reads, (L, S, rows, nb, bn, bfirst) = _all_rows(dst)
    assert (L, S, rows) == (13, 16, 200) and nb == -(-200 // block_rows)
    assert bn.sum() == 200 and (bn[:-1] == min(block_rows, 200)).all()
    assert np.array_equal(bfirst, np.concatenate([[0], np.cumsum(bn)[:-1]]))
    assert np.array_equal(cat["kmer"], f2["kmers"])
    assert np.array_equal(cat["means"], f2["means"].astype(np.float32))
    assert np.array_equal(cat["stds"], f2["stds"].astype(np.float32))
    assert np.array_equal(cat["lens"], f2["lens"])
    assert np.array_equal(cat["signals"], f2["signals"].astype(np.float32))
    assert np.array_equal(cat["labels"], f2["labels"])
    assert info == list(f2["sampleinfo"])
    assert reads == [s.split("\t")[4] for s in f2["sampleinfo"]]


def test_blocks_for_rank_partition(tmp_path):
    dst = os.path.join(str(tmp_path), "f2.dspf")
    featfile.pack_features(os.path.join(GOLDEN, "f2_rows.tsv"), dst, block_rows=9)
    with featfile.FeatureFile(dst) as ff:
        for world in (1, 2, 3, 8, 40):
            cuts = [ff.blocks_for_rank(world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == ff.n_blocks
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
            assert all(a <= b for a, b in cuts)
            if world <= ff.n_blocks // 2:
                rows = [int(ff.block_n[a:b].sum()) for a, b in cuts]
                assert max(rows) - min(rows) <= 2 * 9


def test_reader_thread_yields_same_blocks_as_tsv(tmp_path):
    src = os.path.join(GOLDEN, "f2_rows.tsv")
    dst = os.path.join(str(tmp_path), "f2.dspf")
    featfile.pack_features(src, dst, block_rows=50)
    want = textio.parse_rows(open(src, "rb").read(), 13, 16)
    seen = 0
    for world in (1, 2):
        seen = 0
        for rank in range(world):
            rd = feed.FeatureReader(dst, 13, 16, rank=rank, world=world, nthreads=2, nbuf=2, pinned=False)
            rd.start()
            for blk in rd:
                a, b = blk.first_row, blk.first_row + blk.rows.n
                for k in KEYS:
                    assert np.array_equal(getattr(blk.rows, k), getattr(want, k)[a:b])
                assert [blk.rows.sampleinfo(i) for i in range(blk.rows.n)] == [want.sampleinfo(a + i) for i in range(b - a)]
                # the formatter and call_freq address sampleinfo through (text, row_off, info_len) -- same API
                probs = np.tile(np.array([[0.25, 0.75]], np.float32), (blk.rows.n, 1))
                labels = np.ones(blk.rows.n, np.uint8)
                assert textio.format_calls(blk.rows, probs, labels) == textio.format_calls(want, probs, labels, start=a, stop=b)
                seen += blk.rows.n
                rd.release(blk)
            rd.join()
        assert seen == 200
    with pytest.raises(ValueError):
        feed.FeatureReader(dst, 11, 16, pinned=False)


def test_empty_and_corrupt_files(tmp_path):
    d = str(tmp_path)
    empty_tsv = os.path.join(d, "empty.tsv")
    open(empty_tsv, "w").close()
    dst = os.path.join(d, "empty.dspf")
    assert featfile.pack_features(empty_tsv, dst) == 0
    with featfile.FeatureFile(dst) as ff:
        assert (ff.n_rows, ff.n_blocks) == (0, 0) and ff.blocks_for_rank(2, 1) == (0, 0)
    good = os.path.join(d, "good.dspf")
    featfile.pack_features(os.path.join(GOLDEN, "f2_rows.tsv"), good, block_rows=64)
    blob = open(good, "rb").read()
    bad = os.path.join(d, "bad.dspf")
    open(bad, "wb").write(blob[:len(blob) // 2])  # truncated: index gone
    with pytest.raises(ValueError):
        featfile.FeatureFile(bad)
    open(bad, "wb").write(b"DSPFEAT1" + b"\0" * 56)  # never closed: no index
    with pytest.raises(ValueError):
        featfile.FeatureFile(bad)
    corrupt = bytearray(blob)
    corrupt[64:68] = b"XXXX"  # first block header magic
    open(bad, "wb").write(bytes(corrupt))
    with featfile.FeatureFile(bad) as ff, pytest.raises(ValueError):
        ff.read_block(0)
    with pytest.raises(ValueError):
        featfile.FeatureFile(os.path.join(d, "missing.dspf"))
    w = featfile.FeatureFileWriter(os.path.join(d, "w.dspf"))
    w.close()
    with pytest.raises(ValueError):
        w.add(textio.parse_rows(open(os.path.join(GOLDEN, "f2_rows.tsv"), "rb").read()))


def test_pack_features_cli(tmp_path):
    dst = os.path.join(str(tmp_path), "cli.dspf")
    r = subprocess.run([sys.executable, "-m", "deepsignal_plant_amd.deepsignal_plant", "pack_features", "-i",
                        os.path.join(GOLDEN, "f2_rows.tsv.gz"), "-o", dst, "--block_rows", "100", "-p", "2"],
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "200 rows" in r.stdout
    with featfile.FeatureFile(dst) as ff:
        assert ff.n_rows == 200 and ff.n_blocks == 2
