#!/usr/bin/env python3
"""Host feed of an N-rank call_mods WITHOUT the GPU: every rank parses its byte range of one feature TSV into (unpinned)
SoA buffers with its share of the node's CPUs (dist.threads_per_rank) and formats as many call lines as it parsed rows
-- the two host stages that bracket the forward (reference: _read_features_file, call_modifications.py:55-127; the
per-row strings of _call_mods, :175-188).  Prints one JSON line: per-rank and aggregate rows/s and GB/s of text, next to
what one GPU eats (1.24 M sites/s = 2.6 GB/s of text).  Also `--gz`: the same rows from a foreign
single-stream .gz through the node's shared-memory ring (one inflater per node).
`--parse_on device` (round 4): the rows are parsed on the GPU, so the host's part is what is timed here -- the reader's one
copy + row-start pass into the staging buffer (parse_dev.stage_rows) and the formatting of the call lines, with ONE thread
each; the small per-row arrays the formatter needs (sampleinfo length, k-mer codes: what the device parser copies back) come
from an untimed pre-pass.  The line then also carries the CPU seconds a rank spent (all its threads) and the host threads
a rank needs at the GPU's full rate.
usage: bench_feed.py [--ranks 8] [--rows 400000] [--nproc 10] [--gz] [--affinity slice] [--parse_on device]"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, path, nproc, affinity, q, parse_on="host"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))
    if affinity:
        os.environ["DSP_RANK_AFFINITY"] = affinity
    import numpy as np
    import torch.distributed as dist

    from deepsignal_plant_amd import dist as dd
    from deepsignal_plant_amd import feed, textio
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cpus = dd.pin_rank(rank, world)
    nthreads = dd.threads_per_rank(nproc, world)   # (after pin_rank: the pinned slice is the share)
    ring = None
    first_row, byte_range = 0, None
    if path.endswith(".gz"):
        def gather(obj):
            out = [None] * world
            dist.all_gather_object(out, obj)
            return out
        ring = feed.open_gz_ring(path, rank, world, rank, world, gather)
    else:
        import mmap
        size = os.path.getsize(path)
        with open(path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
            byte_range = dd.byte_range_for_rank(mm, size, world, rank)
    device = parse_on == "device"
    pre = None
    if device:   # untimed: what the device parser would copy back for the formatter (sampleinfo lengths, k-mer codes)
        nthreads = 1
        rd = feed.FeatureReader(path, 13, 16, rank=rank, world=world, nthreads=4, nbuf=2, byte_range=byte_range, pinned=False,
                                gz_ring=None)
        rd.start()
        il, km = [], []
        for blk in rd:
            il.append(blk.rows.info_len.copy()); km.append(blk.rows.kmer.copy())
            rd.release(blk)
        pre = (np.concatenate(il) if il else np.zeros(0, np.uint32), np.concatenate(km) if km else np.zeros((0, 13), np.uint8))
    dist.barrier()
    t0 = time.time()
    c0 = time.process_time()
    t_count = 0.0
    if byte_range is not None and world > 1:   # the counting pass call_mods makes to learn the global row indices
        mine = feed.count_rows_in_range(path, *byte_range, nthreads=nthreads)
        t_count = time.time() - t0
        first_row = dd.exclusive_prefix(dd.all_gather_ints(mine, world), rank)
    reader = feed.FeatureReader(path, 13, 16, rank=rank, world=world, nthreads=nthreads, nbuf=4, first_row=first_row,
                                byte_range=byte_range, pinned=False, gz_ring=ring, device_parse=device)
    reader.start()
    rows = text_bytes = out_bytes = 0
    t_fmt = 0.0
    for blk in reader:
        n = blk.rows.n
        if device:   # (the copies back of the device parser land here)
            blk.rows.info_len[:] = pre[0][rows:rows + n]
            blk.rows.kmer[:] = pre[1][rows:rows + n]
        probs = np.full((n, 2), 0.5, np.float32)
        probs[:, 1] = np.linspace(0.01, 0.99, n, dtype=np.float32)
        probs[:, 0] = 1 - probs[:, 1]
        labels = (probs[:, 1] > 0.5).astype(np.uint8)
        t1 = time.time()
        out_bytes += len(textio.format_calls(blk.rows, probs, labels, nthreads=nthreads))
        t_fmt += time.time() - t1
        rows += n
        text_bytes += int(blk.rows.row_off[-1]) if n and not path.endswith(".gz") else 0
        reader.release(blk)
    dt = time.time() - t0
    cpu = time.process_time() - c0
    dist.barrier()
    if ring is not None:
        ring["ring"].close()
    q.put(dict(rank=rank, rows=rows, seconds=round(dt, 3), threads=nthreads, cpu_seconds=round(cpu, 3), count_seconds=round(t_count, 3), stage_seconds=round(reader.stage_seconds, 3), format_seconds=round(t_fmt, 3), out_bytes=out_bytes,
               cpus=len(cpus) if cpus else None, gz_bytes_in=reader.gz_bytes_in))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rows", type=int, default=400000)
    ap.add_argument("--nproc", type=int, default=10, help="call_mods' --nproc (default 10): threads a rank would like")
    ap.add_argument("--gz", action="store_true", help="feed from a foreign single-stream .gz (gzip -1) of the same rows")
    ap.add_argument("--affinity", default="", choices=["", "slice", "numa"])
    ap.add_argument("--parse_on", default="host", choices=["host", "device"],
                    help="device: time only what the host still does when the GPU parses the rows (stage + format, one thread each)")
    args = ap.parse_args()
    import torch.multiprocessing as mp

    from deepsignal_plant_amd import dist as dd
    work = os.environ.get("DSP_WORK", "/tmp/dsp_pipe")
    os.makedirs(work, exist_ok=True)
    tsv = os.path.join(work, "feat_%d.tsv" % args.rows)
    if not os.path.exists(tsv):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_tsv.py"), tsv, str(args.rows)])
    size = os.path.getsize(tsv)
    path = tsv
    if args.gz:
        path = tsv + ".single.gz"
        if not os.path.exists(path):
            with open(path, "wb") as f:
                subprocess.check_call(["gzip", "-1", "-c", tsv], stdout=f)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    t0 = time.time()
    procs = [ctx.Process(target=_worker, args=(r, args.ranks, port, path, args.nproc, args.affinity, q, args.parse_on)) for r in range(args.ranks)]
    for p in procs:
        p.start()
    res = sorted((q.get() for _ in procs), key=lambda d: d["rank"])
    for p in procs:
        p.join()
    wall = max(r["seconds"] for r in res)
    rows = sum(r["rows"] for r in res)
    assert rows == args.rows, (rows, args.rows)
    cpu = sum(r["cpu_seconds"] for r in res)
    line = {"what": "host feed only (parse + format, no GPU)" if args.parse_on == "host" else
                    "host feed only, rows parsed on the GPU (stage = copy + row starts, format; no GPU here)", "ranks": args.ranks, "host_cpus": dd.available_cpus(),
            "threads_per_rank": res[0]["threads"], "nproc_asked": args.nproc, "affinity": args.affinity or "none",
            "input": "foreign single-stream .gz through the node's shared-memory ring" if args.gz else "plain text, byte ranges",
            "rows": rows, "text_mb": round(size / 1e6, 1), "slowest_rank_s": wall, "process_wall_s": round(time.time() - t0, 2),
            "rows_per_s_all_ranks": round(rows / wall, 1), "text_gb_per_s_all_ranks": round(size / wall / 1e9, 3),
            "rows_per_s_per_rank": round(rows / wall / args.ranks, 1),
            "format_share_of_rank_time": round(sum(r["format_seconds"] for r in res) / sum(r["seconds"] for r in res), 3),
            "one_gpu_needs_rows_per_s": 1.24e6, "ranks_fed_at_full_gpu_rate": round(rows / wall / 1.24e6, 2),
            "count_pass_us_per_row": round(sum(r["count_seconds"] for r in res) / rows * 1e6, 3),
            "stage_us_per_row": round(sum(r["stage_seconds"] for r in res) / rows * 1e6, 3),
            "format_us_per_row": round(sum(r["format_seconds"] for r in res) / rows * 1e6, 3),
            "cpu_seconds_all_ranks": round(cpu, 3), "cpu_us_per_row": round(cpu / rows * 1e6, 3),
            "host_threads_per_rank_at_full_gpu_rate": round(cpu / rows * 1.24e6, 2)}
    if args.gz:
        line["compressed_mb"] = round(os.path.getsize(path) / 1e6, 1)
        line["compressed_bytes_inflated_over_file_size"] = round(sum(r["gz_bytes_in"] for r in res) / os.path.getsize(path), 3)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
