#!/usr/bin/env python3
"""RCCL smoke of the collectives this build uses (one process per GPU): barrier, all_reduce(MAX) of the bench's wall
time, all_gather of row counts / padded probabilities, and the ragged all_to_all_single of the sharded call_freq
(call_mods_freq.DeviceSiteFrequency._exchange).  World size = min(visible GPUs, 2) unless given: on the 1-GPU dev box it
checks world 1, on a node with >= 2 GPUs the real 2-rank exchange over xGMI.  usage: check_rccl.py [world]"""
import os
import socket
import sys


def worker(rank, world, port):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    dist.barrier()
    t = torch.tensor([1.5 + rank], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 1.5 + world - 1
    g = [torch.empty(4, 2, device=dev) for _ in range(world)]
    dist.all_gather(g, torch.full((4, 2), float(rank), device=dev))
    assert all(float(g[r][0, 0]) == r for r in range(world))
    # ragged all_to_all: rank r sends (r + 1) * (d + 1) int64 values to rank d
    send_counts = [(rank + 1) * (d + 1) for d in range(world)]
    recv_counts = [(s + 1) * (rank + 1) for s in range(world)]
    src = torch.cat([torch.full((c,), 100 * rank + d, dtype=torch.int64, device=dev) for d, c in enumerate(send_counts)])
    out = torch.empty(sum(recv_counts), dtype=torch.int64, device=dev)
    dist.all_to_all_single(out, src, recv_counts, send_counts)
    want = torch.cat([torch.full((c,), 100 * s + rank, dtype=torch.int64, device=dev) for s, c in enumerate(recv_counts)])
    assert torch.equal(out, want)
    # the optional final gather of per-site probabilities: ragged, point to point to rank 0 (dist.gather_probs)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from deepsignal_plant_amd import dist as dd
    mine = torch.full((1000 * rank + (5 if rank != 1 else 0), 2), float(rank), device=dev)   # rank 1 has no rows
    got = dd.gather_probs(mine, world)
    if rank == 0:
        assert [int(g.shape[0]) for g in got] == [1000 * r + (5 if r != 1 else 0) for r in range(world)]
        assert all(bool((g == float(r)).all()) for r, g in enumerate(got))
    else:
        assert got is None
    if rank == 0:
        print("rccl ok: world %d, backend %s (barrier, all_reduce, all_gather, ragged all_to_all_single, ragged gather by send/recv)" % (world, dist.get_backend()))
    dist.destroy_process_group()


def main():
    import torch
    import torch.multiprocessing as mp
    ndev = torch.cuda.device_count()  # does not initialise the GPU
    world = int(sys.argv[1]) if len(sys.argv) > 1 else max(1, min(ndev, 2))
    assert 1 <= world <= max(ndev, 1), "one process per GPU: %d GPUs visible" % ndev
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.start_processes(worker, args=(world, port), nprocs=world, join=True, start_method="spawn")


if __name__ == "__main__":
    main()
