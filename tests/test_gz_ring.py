"""CPU: a foreign single-stream .gz (what the reference's `extract --gzip` writes, read back with gzip.open at
call_modifications.py:66-69) read by N ranks of a node: ONE rank inflates, into a shared-memory ring
(csrc/dsp_shmring.cpp, feed.open_gz_ring), and every rank copies its own blocks out.  World 8 and 3 over gloo: total
compressed bytes inflated = 1x the file (it was N x before round 3), every row delivered exactly once, in order, with
its global row index; one rank: the pipelined inflater; truncated / corrupt streams fail loudly."""
import gzip
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _foreign_gz(tmp_path, n_rep=24, unterminated=False):
    data = open(os.path.join(GOLDEN, "f2_rows.tsv"), "rb").read() * n_rep
    if unterminated:
        data = data[:-1]
    p = str(tmp_path / "foreign.tsv.gz")
    with open(p, "wb") as f:   # two members, like `cat a.gz b.gz`: still one sequential stream
        cut = len(data) // 3
        f.write(gzip.compress(data[:cut], 1))
        f.write(gzip.compress(data[cut:], 1))
    return p, data


def _worker(rank, world, port, path, outdir, use_ring):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from deepsignal_plant_amd import feed
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out
    ring = feed.open_gz_ring(path, rank, world, rank, world, gather, block_bytes=300_000) if use_ring else None
    assert (ring is not None) == use_ring
    reader = feed.FeatureReader(path, 13, 16, rank=rank, world=world, nthreads=2, nbuf=2, block_bytes=300_000,
                                pinned=False, gz_ring=ring)
    reader.start()
    firsts, counts, infos = [], [], []
    for blk in reader:
        firsts.append(blk.first_row)
        counts.append(blk.rows.n)
        infos += [blk.rows.sampleinfo(i) for i in range(blk.rows.n)]
        reader.release(blk)
    np.savez(os.path.join(outdir, "r%d.npz" % rank), firsts=np.array(firsts, np.int64), counts=np.array(counts, np.int64),
             infos=np.array(infos), bytes_in=reader.gz_bytes_in)
    dist.barrier()
    if ring is not None:
        ring["ring"].close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,use_ring,unterminated,n_rep", [(8, True, False, 24), (3, True, True, 24), (3, False, False, 24),
                                                               (4, True, False, 1)])
def test_foreign_gz_is_inflated_once_per_node(tmp_path, world, use_ring, unterminated, n_rep):
    import torch.multiprocessing as mp
    path, data = _foreign_gz(tmp_path, n_rep=n_rep, unterminated=unterminated)   # n_rep 1: fewer blocks than ranks
    mp.start_processes(_worker, args=(world, _free_port(), path, str(tmp_path), use_ring), nprocs=world, join=True,
                       start_method="spawn")
    want = ["\t".join(l.split("\t")[:6]) for l in data.decode().splitlines()]
    blocks, total_in = [], 0
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        total_in += int(d["bytes_in"])
        pos = 0
        for f, c in zip(d["firsts"], d["counts"]):
            blocks.append((int(f), r, list(d["infos"][pos:pos + int(c)])))
            pos += int(c)
    blocks.sort()
    # block i went to rank i % world, carries the global index of its first row, and the blocks tile the file in order
    got = []
    for i, (first, r, infos) in enumerate(blocks):
        assert r == i % world and first == len(got)
        got += infos
    assert got == want and (len(blocks) > 2 * world or n_rep == 1)
    size = os.path.getsize(path)
    assert total_in == (size if use_ring else world * size)   # the point: one inflater per node
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("dsp_gz_")]   # the ring is unlinked by its creator


def test_single_rank_reads_a_foreign_gz_through_the_pipelined_inflater(tmp_path):
    from deepsignal_plant_amd import feed, textio
    path, data = _foreign_gz(tmp_path, n_rep=6)
    reader = feed.FeatureReader(path, 13, 16, nthreads=3, nbuf=3, block_bytes=200_000, pinned=False)
    reader.start()
    n, means = 0, []
    for blk in reader:
        assert blk.first_row == n
        n += blk.rows.n
        means.append(blk.rows.means.copy())
        reader.release(blk)
    ref = textio.parse_rows(data, 13, 16)
    assert n == ref.n and np.array_equal(np.concatenate(means), ref.means)
    assert reader.gz_bytes_in == os.path.getsize(path)


@pytest.mark.parametrize("how", ["truncated", "corrupt", "garbage_tail", "zero_padded", "empty"])
def test_damaged_gzip_streams_fail_loudly(tmp_path, how):
    """gzread reports a stream cut in the middle as a clean end of file; the reference's gzip.open raises EOFError.  A
    half-copied feature file must not produce a partial result with exit code 0."""
    from deepsignal_plant_amd import feed, gzio
    path, data = _foreign_gz(tmp_path, n_rep=2)
    raw = open(path, "rb").read()
    p = str(tmp_path / (how + ".tsv.gz"))
    ok = False
    if how == "truncated":
        raw = raw[:len(raw) // 2]
    elif how == "corrupt":
        raw = raw[:5000] + bytes([raw[5000] ^ 0x5a]) + raw[5001:]
    elif how == "garbage_tail":
        raw = raw + b"this is not gzip"
    elif how == "zero_padded":
        raw, ok = raw + bytes(512), True       # tape-style zero padding: gzip.open accepts it too
    elif how == "empty":
        raw, data, ok = b"", b"", True
    open(p, "wb").write(raw)
    if ok:
        assert gzip.open(p, "rb").read() == data
    else:
        with pytest.raises((EOFError, OSError, gzip.BadGzipFile)):
            gzip.open(p, "rb").read()
    st = gzio.GzStream(p)
    buf = np.empty(len(data) + 100, np.uint8)
    if ok:
        n = st.readinto(buf)
        assert buf[:n].tobytes() == data and st.readinto(buf) == 0
    else:
        with pytest.raises(ValueError, match="truncated gzip stream" if how == "truncated" else "corrupt gzip stream"):
            got = 0
            while True:
                k = st.readinto(buf, got)
                if k == 0:
                    break
                got += k
    st.close()
    # and through the reader: the error reaches the consumer
    reader = feed.FeatureReader(p, 13, 16, nthreads=2, block_bytes=100_000, pinned=False)
    reader.start()
    if ok:
        n = 0
        for b in reader:
            n += b.rows.n
            reader.release(b)
        assert n == data.count(b"\n")
    else:
        # (a flipped bit garbles the text long before the member's CRC is reached: the parser may object first)
        with pytest.raises(ValueError, match="gzip stream|malformed feature row"):
            for b in reader:
                reader.release(b)


@pytest.mark.parametrize("is_gzip", [False, True])
def test_interleaved_part_files_merge_back_into_input_order(tmp_path, is_gzip):
    """With a foreign .gz the ranks own blocks round-robin; each rank's writer records where every block's calls end in
    its part file (whole BGZF members under --gzip: BgzfWriter.end_block) and rank 0 interleaves the pieces again
    (call_modifications._merge_parts): the result is the input-order file, one BGZF end-of-file member at the end."""
    from deepsignal_plant_amd import call_modifications as cm
    from deepsignal_plant_amd import gzio
    world = 3
    rng = np.random.default_rng(5)
    blocks = [("block %d " % i).encode() * int(rng.integers(1, 9000)) + b"\n" for i in range(11)]  # rank 2 gets one block less
    blocks[4] = b""                                                                                 # a block without rows
    out = str(tmp_path / ("calls.tsv" + (".gz" if is_gzip else "")))
    for r in range(world):
        part = "%s.part%05d" % (out, r)
        ends, pos = [], 0
        wf = gzio.open_write(part, is_gzip, nthreads=2)
        with wf:
            for b in blocks[r::world]:
                wf.write(b)
                if is_gzip:
                    wf.end_block()
                else:
                    pos += len(b)
                    ends.append(pos)
        if is_gzip:
            ends = list(wf.block_ends)
        assert len(ends) == len(blocks[r::world])
        np.asarray(ends, np.int64).tofile(part + ".blocks")
    cm._merge_parts(out, world, True)
    got = gzip.open(out, "rb").read() if is_gzip else open(out, "rb").read()
    assert got == b"".join(blocks)
    assert sorted(os.listdir(str(tmp_path))) == [os.path.basename(out)]
    if is_gzip:
        bz = gzio.BgzfFile(out)
        assert bz.ok and int(bz.isize[bz.n_members - 1]) == 0 and list(bz.isize[:bz.n_members - 1]).count(0) == 0


def test_stale_piece_tables_do_not_choose_the_merge(tmp_path):
    """an aborted run on a foreign .gz leaves `<part>.blocks` files behind.  A later contiguous run with the same
    -o must not interleave its part files by those stale tables: the merge is told what THIS run did, the tables of an
    earlier one are removed when a rank starts, and a table that does not describe its part file is an error."""
    from deepsignal_plant_amd import call_modifications as cm
    world = 2
    out = str(tmp_path / "calls.tsv")
    parts = ["%s.part%05d" % (out, r) for r in range(world)]
    for r, p in enumerate(parts):
        open(p, "wb").write(b"stale part %d\n" % r)
        np.asarray([3, 7], np.int64).tofile(p + ".blocks")      # what the aborted run left
    for p in parts:                                              # every rank clears its own leftovers at start ...
        cm._remove_stale_parts(p, world)
    assert os.listdir(str(tmp_path)) == []
    for r, p in enumerate(parts):                                # ... and a contiguous run writes its parts
        open(p, "wb").write(b"rank %d rows\n" % r * 5)
    np.asarray([3, 7], np.int64).tofile(parts[0] + ".blocks")    # (even a table that survived is not consulted)
    np.asarray([3, 7], np.int64).tofile(parts[1] + ".blocks")
    cm._merge_parts(out, world, False)
    assert open(out, "rb").read() == b"rank 0 rows\n" * 5 + b"rank 1 rows\n" * 5
    # an interleaved merge checks every table against its part file
    for r, p in enumerate(parts):
        open(p, "wb").write(b"rank %d rows\n" % r * 5)
        np.asarray([3, 7], np.int64).tofile(p + ".blocks")
    with pytest.raises(RuntimeError, match="does not describe"):
        cm._merge_parts(out, world, True)
