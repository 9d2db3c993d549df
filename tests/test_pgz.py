"""CPU: the parallel inflater of ONE gzip stream (csrc/dsp_pgz.cpp; what reads a feature file written by the reference's
`extract --gzip`, call_modifications.py:66-69, at more than zlib's single-thread rate).  Differential against Python's
gzip on everything a gzip writer can produce -- compression levels, many members, zero padding, header fields, fixed and
stored blocks, binary data (no block start is ever found: the chunks fall back to one true decoder), tiny chunks that
force hundreds of chunk boundaries and windows shorter than 32 KiB -- and the same loud errors as the sequential reader
on truncated and corrupt streams (a damaged file must never yield a short result silently)."""
import gzip
import os
import zlib

import numpy as np
import pytest

from deepsignal_plant_amd import gzio
from tests.helpers import GOLDEN


def _text(n_rep):
    return open(os.path.join(GOLDEN, "f2_rows.tsv"), "rb").read() * n_rep


def _read_all(path, nthreads, chunk, bufsize=1_000_003):
    st = gzio.PgzStream(path, nthreads, chunk)
    buf = np.empty(bufsize, np.uint8)
    parts = []
    try:
        while True:
            k = st.readinto(buf)
            if k == 0:
                break
            parts.append(buf[:k].tobytes())
        stats = st.stats()
        assert st.bytes_in() == os.path.getsize(path)
    finally:
        st.close()
    return b"".join(parts), stats


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("nthreads,chunk", [(1, 0), (3, 65536), (8, 300_000), (5, 1 << 20)])
def test_parallel_inflate_equals_gzip(tmp_path, level, nthreads, chunk):
    data = _text(14)                                   # 5.8 MB of feature rows
    p = str(tmp_path / "x.gz")
    open(p, "wb").write(gzip.compress(data, level))
    got, (rounds, dropped) = _read_all(p, nthreads, chunk)
    assert got == data
    assert rounds >= 1
    if nthreads > 1 and chunk and os.path.getsize(p) > 2 * nthreads * chunk:
        assert rounds >= 2
    assert dropped <= 1                                # on text a chunk start is practically never a false positive


def test_members_padding_header_fields_and_block_kinds(tmp_path):
    rng = np.random.default_rng(5)
    text = _text(3)
    noise = rng.integers(0, 256, 700_000, dtype=np.uint8).tobytes()          # incompressible: stored blocks
    pieces = [text[:900_000], b"", noise, text[900_000:], b"A", bytes(70_000), text[:5000]]
    raw = b""
    for i, pc in enumerate(pieces):
        c = zlib.compressobj([1, 6, 9][i % 3], zlib.DEFLATED, 31)
        raw += c.compress(pc) + c.flush()
        if i == 3:
            raw += bytes(1000)                                                # zero padding between members
    # a member with FNAME / FCOMMENT / FEXTRA / FHCRC header fields, and a fixed-Huffman member (Z_FIXED)
    hdr = bytes([0x1f, 0x8b, 8, 2 | 4 | 8 | 16, 0, 0, 0, 0, 0, 3]) + bytes([3, 0]) + b"xyz" + b"name.tsv\0" + b"a comment\0"
    hdr += (zlib.crc32(hdr) & 0xffff).to_bytes(2, "little")
    body = zlib.compressobj(6, zlib.DEFLATED, -15)
    tail = text[:40_000]
    raw += hdr + body.compress(tail) + body.flush() + zlib.crc32(tail).to_bytes(4, "little") + (len(tail) & 0xffffffff).to_bytes(4, "little")
    fx = zlib.compressobj(9, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)
    raw += fx.compress(text[:30_000]) + fx.flush()
    raw += bytes(300)
    want = b"".join(pieces) + tail + text[:30_000]
    p = str(tmp_path / "m.gz")
    open(p, "wb").write(raw)
    assert gzip.open(p, "rb").read() == want                                  # the file is what Python's gzip reads
    for nthreads, chunk in ((1, 0), (4, 65536), (7, 200_000)):
        got, _ = _read_all(p, nthreads, chunk)
        assert got == want, (nthreads, chunk)


def test_non_ascii_streams_are_still_decoded_exactly(tmp_path):
    """block starts are recognised by ASCII literals; data that is not text never offers one, and every round is decoded
    by its first chunk alone: slow, exact"""
    rng = np.random.default_rng(7)
    data = (rng.integers(0, 4, 3_000_000, dtype=np.uint8) * 60 + 130).astype(np.uint8).tobytes()   # compressible, all >= 128
    p = str(tmp_path / "b.gz")
    open(p, "wb").write(gzip.compress(data, 6))
    got, (rounds, dropped) = _read_all(p, 4, 65536)
    assert got == data


@pytest.mark.parametrize("how", ["truncated", "truncated_at_trailer", "flipped", "bad_crc", "bad_isize", "garbage_tail"])
def test_damaged_streams_fail_like_the_sequential_reader(tmp_path, how):
    data = _text(10)
    raw = bytearray(gzip.compress(data, 6))
    if how == "truncated":
        raw = raw[:len(raw) * 2 // 3]
    elif how == "truncated_at_trailer":
        raw = raw[:-5]
    elif how == "flipped":
        raw[len(raw) // 2] ^= 0x10
    elif how == "bad_crc":
        raw[-8] ^= 1
    elif how == "bad_isize":
        raw[-1] ^= 1
    elif how == "garbage_tail":
        raw += b"not a gzip member"
    p = str(tmp_path / (how + ".gz"))
    open(p, "wb").write(bytes(raw))
    with pytest.raises((EOFError, OSError, zlib.error, gzip.BadGzipFile)):
        gzip.open(p, "rb").read()
    for nthreads, chunk in ((1, 0), (4, 100_000)):
        with pytest.raises(ValueError, match="gzip stream"):
            _read_all(p, nthreads, chunk)
    if how in ("truncated", "truncated_at_trailer"):
        with pytest.raises(ValueError, match="truncated gzip stream"):
            _read_all(p, 3, 150_000)


def test_reader_uses_the_parallel_inflater_and_yields_the_same_rows(tmp_path, monkeypatch):
    """feed.FeatureReader on a foreign .gz: with host threads to spare the stream goes through the parallel inflater
    (forced here for a small file), with DSP_GZ_SEQUENTIAL=1 through zlib: same blocks of rows"""
    from deepsignal_plant_amd import feed, textio
    data = _text(12)
    p = str(tmp_path / "f.tsv.gz")
    open(p, "wb").write(gzip.compress(data, 9))
    ref = textio.parse_rows(data, 13, 16)
    monkeypatch.setattr(gzio, "PGZ_MIN_BYTES", 0)
    for seq in (False, True):
        if seq:
            monkeypatch.setenv("DSP_GZ_SEQUENTIAL", "1")
        reader = feed.FeatureReader(p, 13, 16, nthreads=4, nbuf=3, block_bytes=400_000, pinned=False)
        reader.start()
        n, sig = 0, []
        for blk in reader:
            assert blk.first_row == n
            n += blk.rows.n
            sig.append(blk.rows.signals.copy())
            reader.release(blk)
        assert n == ref.n and np.array_equal(np.concatenate(sig), ref.signals)
        assert reader.gz_bytes_in == os.path.getsize(p)
        assert reader.gz_parallel == (not seq)


def test_streams_without_recognisable_block_starts_fall_back_to_one_decoder(tmp_path):
    """incompressible high-bit data: every literal fails the ASCII check, so no chunk but the first ever finds a start;
    after two such rounds the inflater stops sending threads to search (they would only make the true decoder wait) and
    still delivers the exact bytes"""
    rng = np.random.default_rng(3)
    data = rng.integers(128, 256, 1_500_000, dtype=np.uint16).astype(np.uint8).tobytes()
    p = str(tmp_path / "hi.gz")
    open(p, "wb").write(gzip.compress(data, 6))
    got, (rounds, dropped) = _read_all(p, 4, 65536)
    assert got == data
    assert rounds >= 6 and dropped == 6          # two rounds of 4 chunks lost their 3 searchers, then one-chunk rounds
