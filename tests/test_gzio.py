"""CPU: the native gzip layer (csrc/dsp_gz.cpp, gzio.py): what --gzip writes is a BGZF chain that Python's gzip (the
reference's reader, call_modifications.py:66-69) reads back byte for byte; BGZF files are read by member ranges on N
threads, foreign .gz files through the streaming inflater; rank splits cover every row exactly once."""
import gzip
import os

import numpy as np
import pytest

from deepsignal_plant_amd import feed, gzio, textio
from tests.helpers import GOLDEN


def _text(n_rep=40):
    return open(os.path.join(GOLDEN, "f2_rows.tsv"), "rb").read() * n_rep   # 8,000 rows, 16.7 MB


def test_bgzf_writer_output_is_plain_gzip_and_indexable(tmp_path):
    data = _text(8)
    p = str(tmp_path / "x.tsv.gz")
    with gzio.BgzfWriter(p, nthreads=4, chunk=1 << 20) as w:
        for a in range(0, len(data), 300_001):   # ragged writes
            w.write(data[a:a + 300_001])
    assert gzip.open(p, "rb").read() == data                      # the reference's reader
    bz = gzio.BgzfFile(p)
    assert bz.ok and bz.n_members >= len(data) // 0xff00 and int(bz.text_off[-1]) == len(data)
    assert int(bz.isize[bz.n_members - 1]) == 0                   # the empty end-of-file member
    out, n = bz.inflate(0, bz.n_members, nthreads=5)
    assert n == len(data) and out[:n].tobytes() == data
    out, n = bz.inflate(3, 9, nthreads=2)
    assert out[:n].tobytes() == data[int(bz.text_off[3]):int(bz.text_off[9])]
    # empty input: just the EOF member
    p2 = str(tmp_path / "empty.gz")
    gzio.BgzfWriter(p2).close()
    assert gzip.open(p2, "rb").read() == b"" and os.path.getsize(p2) == 28
    # a corrupted member is refused (zlib checks the CRC of every member)
    raw = bytearray(open(p, "rb").read())
    raw[int(bz.off[2]) + 40] ^= 0x55
    p3 = str(tmp_path / "bad.gz")
    open(p3, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="corrupt gzip member"):
        gzio.BgzfFile(p3).inflate(0, 5)


def _read_all(path, world, rank, first_row=0, block_bytes=700_000):
    r = feed.FeatureReader(path, 13, 16, rank=rank, world=world, nthreads=3, nbuf=2, block_bytes=block_bytes,
                           first_row=first_row, pinned=False)
    r.start()
    rows, firsts = [], []
    for blk in r:
        firsts.append(blk.first_row)
        rows += [blk.rows.sampleinfo(i) for i in range(blk.rows.n)]
        r.release(blk)
    return rows, firsts


@pytest.mark.parametrize("kind", ["bgzf", "single_member", "no_trailing_newline"])
def test_gz_feature_files_are_read_completely_by_any_number_of_ranks(tmp_path, kind):
    data = _text(6)
    if kind == "no_trailing_newline":
        data = data[:-1]
    want = ["\t".join(l.split("\t")[:6]) for l in data.decode().splitlines()]
    p = str(tmp_path / "f.tsv.gz")
    if kind == "single_member":
        with gzip.open(p, "wb", compresslevel=1) as f:
            f.write(data)
    else:
        with gzio.BgzfWriter(p, nthreads=3) as w:
            w.write(data)
    for world in (1, 2, 3):
        got, prefix = [], 0
        for rank in range(world):
            if kind == "single_member":
                assert feed.count_rows_bgzf(p, world, rank) is None
                rows, firsts = _read_all(p, world, rank)        # every rank inflates everything, keeps its blocks
                got.append((firsts, rows))
            else:
                mine = feed.count_rows_bgzf(p, world, rank, nthreads=2, block_bytes=500_000)
                rows, firsts = _read_all(p, world, rank, first_row=prefix)
                assert len(rows) == mine + (1 if kind == "no_trailing_newline" and rank == world - 1 else 0)
                assert not firsts or firsts[0] == prefix
                prefix += mine
                got.append((firsts, rows))
        if kind == "single_member":  # blocks interleave over ranks: order them by their first row
            allfirst = sorted(f for firsts, _ in got for f in firsts)
            merged = {}
            for firsts, rows in got:
                pos = 0
                for f in firsts:
                    nxt = [x for x in allfirst if x > f]
                    size = (nxt[0] if nxt else len(want)) - f
                    merged[f] = rows[pos:pos + size]
                    pos += size
            flat = [r for f in sorted(merged) for r in merged[f]]
            assert flat == want
        else:
            assert [r for _, rows in got for r in rows] == want


def test_streaming_inflater_reads_concatenated_members(tmp_path):
    a, b = _text(1), _text(2)
    p = str(tmp_path / "cat.gz")
    with open(p, "wb") as f:
        f.write(gzip.compress(a))
        f.write(gzip.compress(b))
    st = gzio.GzStream(p)
    buf = np.empty(len(a) + len(b) + 10, np.uint8)
    n = st.readinto(buf)
    assert buf[:n].tobytes() == a + b and st.readinto(buf) == 0
    st.close()
    assert textio.count_newlines(np.frombuffer(a, np.uint8)) == a.count(b"\n")
    assert textio.count_newlines(np.frombuffer(a[:-1], np.uint8)) == a.count(b"\n") - 1


def test_background_file_writer_keeps_order_and_surfaces_errors(tmp_path):
    """the plain-text writer of `extract`: buffers (bytes, memoryviews, arrays) appended by its own thread, in order"""
    p = str(tmp_path / "o.tsv")
    parts = [b"abc\n" * 1000, memoryview(b"defg\n" * 7), np.frombuffer(b"xyz\n" * 3, np.uint8), b""]
    with gzio.open_write(p, False, background=True) as wf:
        assert isinstance(wf, gzio.BackgroundFileWriter)
        for x in parts:
            wf.write(x)
    assert open(p, "rb").read() == b"".join(bytes(x) for x in parts)
    done = []
    with gzio.open_write(p, False, background=True) as wf:   # lists of buffers, and the callback that frees them
        wf.write([b"12", memoryview(b"34")], on_done=lambda: done.append(1))
        wf.write(b"5", on_done=lambda: done.append(2))
    assert open(p, "rb").read() == b"12345" and done == [1, 2]
    wf = gzio.open_write(str(tmp_path / "e.tsv"), False, background=True)
    wf.write("not bytes")            # the worker's TypeError must come back to the caller, not vanish
    with pytest.raises(TypeError):
        wf.close()


def test_bgzf_writer_takes_lists_of_buffers_without_copying_them_through_the_carry(tmp_path):
    """`extract --gzip`: the formatter's parts are deflated as they are; a part ends its own member, the file is still
    one BGZF chain that any gzip reader reads, and on_done fires after the parts are in the file"""
    p = str(tmp_path / "parts.tsv.gz")
    a, b, c = _text(2), _text(1)[:70001], b"tail\n"
    done = []
    with gzio.open_write(p, True, nthreads=3) as wf:
        wf.write(b"head\n")
        wf.write([memoryview(a), np.frombuffer(b, np.uint8), b""], on_done=lambda: done.append(len(done)))
        wf.write(c)
        wf.write([c, c], on_done=lambda: done.append(len(done)))
        with pytest.raises(ValueError):
            wf.write(c, on_done=lambda: None)
    want = b"head\n" + a + b + c + c + c
    assert gzip.open(p, "rb").read() == want and done == [0, 1]
    assert gzio.BgzfFile(p).ok


@pytest.mark.parametrize("gz", [False, True])
def test_reader_on_degenerate_files(tmp_path, gz):
    """what a user can hand to call_mods -i: an empty file, one row without a newline, CRLF ends, extra columns (ignored
    like words[11] ignores them, call_modifications.py:75-92) -- and a blank line, which the reference dies on with an
    IndexError in a worker (words[4]) and this reader refuses with an IndexError that names the row"""
    rows = open(os.path.join(GOLDEN, "f2_rows.tsv"), "rb").read().split(b"\n")[:5]
    cases = {"empty": (b"", 0), "one_no_newline": (rows[0], 1), "crlf": (b"\r\n".join(rows[:3]) + b"\r\n", 3),
             "extra_col": (rows[0] + b"\textra\n" + rows[1] + b"\n", 2), "blank_line": (rows[0] + b"\n\n" + rows[1] + b"\n", None)}
    for name, (data, want) in cases.items():
        p = str(tmp_path / ("%s.tsv%s" % (name, ".gz" if gz else "")))
        open(p, "wb").write(gzip.compress(data) if gz else data)
        reader = feed.FeatureReader(p, 13, 16, nthreads=2, nbuf=2, pinned=False)
        reader.start()
        n = 0
        if want is None:
            with pytest.raises(IndexError, match="malformed feature row 1"):   # (the reference's exception type, the row number besides)
                for blk in reader:
                    reader.release(blk)
            continue
        for blk in reader:
            assert blk.first_row == n
            n += blk.rows.n
            reader.release(blk)
        assert n == want, name


def test_bgzf_members_carry_their_row_counts_and_the_counting_pass_uses_them(tmp_path, monkeypatch):
    """every member this build writes records its newline count in the gzip header's MTIME field under XFL = 'R' (fields no
    reader interprets: Python's gzip and the BGZF member walk read the file as before); a rank then learns the rows of its
    member range from the headers instead of inflating it -- same counts as the inflate pass for every world size, and a
    BGZF file without the marks (another writer's) still takes the inflate pass"""
    data = _text(5)
    p = str(tmp_path / "rows.tsv.gz")
    with gzio.open_write(p, True, nthreads=3) as wf:
        wf.write(data[:700_001])
        wf.write([memoryview(data[700_001:1_500_000]), data[1_500_000:]])
    assert gzip.open(p, "rb").read() == data
    bz = gzio.BgzfFile(p)
    assert bz.ok and bz.n_members > 20 and int(bz.rows[bz.n_members - 1]) == 0   # the empty end-of-file member
    for m in range(bz.n_members - 1):
        buf, n = bz.inflate(m, m + 1)
        assert int(bz.rows[m]) == bytes(buf[:n]).count(b"\n"), m
    for world in (1, 2, 3, 4):
        by_header = [feed.count_rows_bgzf(p, world, r, nthreads=2) for r in range(world)]
        monkeypatch.setenv("DSP_BGZF_COUNT_BY_INFLATE", "1")
        by_inflate = [feed.count_rows_bgzf(p, world, r, nthreads=2) for r in range(world)]
        monkeypatch.delenv("DSP_BGZF_COUNT_BY_INFLATE")
        assert by_header == by_inflate and sum(by_header) == data.count(b"\n")
    raw = bytearray(open(p, "rb").read())                                   # the same file as another BGZF writer leaves it
    for off in gzio.BgzfFile(p).off[:-1]:
        raw[int(off) + 4:int(off) + 9] = bytes(5)
    q = str(tmp_path / "foreign_bgzf.tsv.gz")
    open(q, "wb").write(bytes(raw))
    fz = gzio.BgzfFile(q)
    assert fz.ok and bool((fz.rows[:fz.n_members - 1] < 0).all())
    assert [feed.count_rows_bgzf(q, 3, r, nthreads=2) for r in range(3)] == [feed.count_rows_bgzf(p, 3, r, nthreads=2) for r in range(3)]
    assert gzip.open(q, "rb").read() == data
