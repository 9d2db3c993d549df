"""Range-sharding of independent sites over the GPUs of one node.

Every site (feature row) is independent (the forward has no cross-row term, models.py:178-240), so the N
ranks take contiguous ranges with NO data-path collective.  The only exchanges are (1) an all_gather of one
integer per rank (rows in my byte range) so that each rank knows the GLOBAL index of its first row -- the
in-kernel Philox initial states are keyed by it, which makes results independent of N -- and (2) the
optional final gather of per-site probabilities.  Backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in
the CPU tests."""
from __future__ import annotations

import os


def env_world():
    """(rank, local_rank, world) from the torch.distributed.run environment (1 process per GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def forced():
    """DSP_FORCE_DIST=1: take the distributed branches (process group over RCCL, collectives, exchange) even with ONE
    rank -- so that the RCCL code paths of bench.py / call_mods / the sharded call_freq execute on a 1-GPU box."""
    return os.environ.get("DSP_FORCE_DIST") == "1"


def collective(world):
    """True when the collectives of the path must run: several ranks, or a forced one-rank process group"""
    if world > 1:
        return True
    if not forced():
        return False
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def init_process_group(world, rank, dev_index, ndev):
    """One process per GPU: RCCL ("nccl" on ROCm) bound to this rank's device; when fewer GPUs than ranks are visible (the
    1-GPU dev box) the ranks share GPUs and the control plane runs over gloo.  With DSP_FORCE_DIST=1 a lone rank builds
    its own one-rank RCCL group.  Returns the backend name, or None when no group is needed."""
    if world == 1 and not forced():
        return None
    import socket

    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world == 1 and "MASTER_PORT" not in os.environ:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        s.close()
    if dist.is_initialized():
        return dist.get_backend()
    torch.cuda.set_device(dev_index)
    if ndev >= world:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist.get_backend()


def visible_gpu_count():
    """GPUs this process would see, WITHOUT loading the HIP runtime (the launchers decide how many ranks to start before
    anything may touch the GPU: a process that initialised the GPU must not start replacing itself).  The device
    filters of the ROCm stack first (HIP_VISIBLE_DEVICES on top of ROCR_VISIBLE_DEVICES; CUDA_VISIBLE_DEVICES is
    honoured by HIP too), else the KFD topology (nodes with SIMDs are GPUs), else a fresh child process that asks torch."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [x for x in v.split(",") if x.strip() != ""]
            if v.strip() in ("", "-1"):
                return 0
            return len(ids)
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        if n > 0:
            return n
    except (OSError, ValueError):
        pass
    import subprocess
    import sys
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                             text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def available_cpus():
    """host threads this process may use: CPU affinity, capped by the cgroup CPU quota"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


_PINNED_SHARE = None   # set by pin_rank: the process affinity IS this rank's share already


def threads_per_rank(nproc, local_world=None):
    """Host threads (parser / formatter / deflate) of ONE rank: --nproc, capped by this rank's share of the node's CPUs
    (available CPUs // ranks on the node) -- 8 ranks must not each start --nproc threads on a node that has fewer than
    8 x --nproc cores.  After pin_rank has narrowed the affinity to the rank's own CPUs the share is that set, not a
    local_world-th of it (128 CPUs and 8 pinned ranks gave 2 threads instead of 16).  DSP_THREADS_PER_RANK
    overrides."""
    env = os.environ.get("DSP_THREADS_PER_RANK")
    if env:
        return max(1, int(env))
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if _PINNED_SHARE is not None:
        share = max(1, min(available_cpus(), _PINNED_SHARE))
    else:
        share = max(1, available_cpus() // max(1, local_world))
    return max(1, min(nproc if nproc and nproc > 0 else 1, share))


def spare_cpus(local_world):
    """CPUs a node-wide helper of this rank (the one inflater of a foreign .gz) may use next to the node's ranks: what
    is allowed minus one per rank -- of the NODE's CPUs when this rank is not pinned, of its own slice when it is"""
    if _PINNED_SHARE is not None:
        return max(1, available_cpus() - 1)
    return max(1, available_cpus() - max(1, local_world))


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


_SYSFS = os.environ.get("DSP_SYSFS_ROOT", "/sys")   # (the tests point this at a faked 2-socket 8-GPU tree)


def pci_bdf(domain, bus, device, function=0):
    """sysfs name of a PCI function from the integers torch's device properties carry (pci_domain_id, pci_bus_id,
    pci_device_id -- in torch 2.10+rocm7 `pci_bus_id` is the bus NUMBER, not a string: ADVICE r4)"""
    return "%04x:%02x:%02x.%x" % (int(domain), int(bus), int(device), int(function))


def _numa_node_cpus(bdf):
    """(NUMA node of the PCI device named like /sys/bus/pci/devices/<bdf>, the node's CPU list) or (None, None) when the
    kernel reports no node for it (numa_node = -1: one memory domain) or the device is not there"""
    if not isinstance(bdf, str) or not bdf:
        return None, None
    try:
        node = int(open(os.path.join(_SYSFS, "bus/pci/devices", bdf.lower(), "numa_node")).read().strip())
        if node < 0:
            return None, None
        return node, _cpulist(open(os.path.join(_SYSFS, "devices/system/node/node%d/cpulist" % node)).read())
    except (OSError, ValueError):
        return None, None


def affinity_mode(local_world):
    """DSP_RANK_AFFINITY = numa | slice | off.  Default: numa when several ranks share the node (a 2-socket 8-GPU node
    puts half the ranks' page-locked slots and staging threads on the wrong socket otherwise), off for
    a lone rank."""
    mode = os.environ.get("DSP_RANK_AFFINITY", "").strip().lower()
    if mode in ("off", "none", "0", "no"):
        return ""
    if mode in ("numa", "slice"):
        return mode
    return "numa" if local_world > 1 else ""


def local_gpu_bdfs(local_world, ndev, bdf_of):
    """PCI names of the GPUs of ALL local ranks as this rank can see them: local rank r runs on visible device r % ndev
    (call_modifications.py:523-529 maps processes to devices the same way).  None when the launcher narrowed this rank's
    view to fewer devices than ranks AND ranks do not simply share them (then only the own GPU is known)."""
    try:
        return [bdf_of(r % ndev) for r in range(local_world)]
    except Exception:
        return None


def pin_rank(local_rank, local_world, pci_bus_id=None, peer_bus_ids=None):
    """CPU placement of a rank (affinity_mode: numa by default with several ranks; DSP_RANK_AFFINITY=off leaves the
    scheduler alone).  Call it BEFORE the page-locked slots are allocated and the staging / writer threads start: both
    then live next to the rank's GPU.
    slice: the local_rank-th of local_world equal slices of the allowed CPUs.
    numa: the CPUs of the NUMA node the rank's GPU hangs off (/sys/bus/pci/devices/<bdf>/numa_node; pci_bus_id is that
    sysfs name, dist.pci_bdf / _native.device_pci_bdf); when the caller names the GPUs of ALL local ranks (peer_bus_ids[r] =
    name of local rank r's GPU) the node's CPUs are split in equal slices among the ranks whose GPUs share that node;
    without that list every rank of the node takes the whole node (and threads_per_rank divides by the ranks per node as
    if unpinned).  A GPU whose node the kernel does not report (one memory domain): with DSP_RANK_AFFINITY=numa asked for
    explicitly the rank falls back to slice (said on stderr); by default it stays unpinned.
    Returns the CPU list set, or None.  A rank whose affinity became its own share records it for threads_per_rank."""
    global _PINNED_SHARE
    mode = affinity_mode(local_world)
    if not mode or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    cpus, own_share = None, False
    if mode == "numa":
        node, node_cpus = _numa_node_cpus(pci_bus_id)
        cand = [c for c in (node_cpus or []) if c in allowed]
        if cand:
            cpus = cand
            if peer_bus_ids:   # my slice of the node's CPUs among the local ranks on the same node
                peers = [r for r, b in enumerate(peer_bus_ids) if _numa_node_cpus(b)[0] == node]
                if local_rank in peers and len(cand) >= len(peers):
                    k = len(cand) // len(peers)
                    i = peers.index(local_rank)
                    cpus, own_share = cand[i * k:(i + 1) * k], True
        elif os.environ.get("DSP_RANK_AFFINITY", "").strip().lower() == "numa":
            import sys
            sys.stderr.write("[dist] DSP_RANK_AFFINITY=numa: no NUMA node for GPU %r of local rank %d under %s "
                             "(numa_node < 0 or unreadable): taking an equal slice of the allowed CPUs instead\n"
                             % (pci_bus_id, local_rank, _SYSFS))
        else:
            # numa by DEFAULT and the kernel reports one memory domain (or an outer launcher hid the device): nothing to be
            # next to -- narrowing every rank to a hard 1 / local_world slice would only cap its threads: unpinned
            return None
    if cpus is None:
        k = max(1, len(allowed) // max(1, local_world))
        cpus = allowed[local_rank * k:(local_rank + 1) * k]
        own_share = bool(cpus)
        cpus = cpus or allowed
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    _PINNED_SHARE = len(cpus) if own_share else None
    return cpus


_PLACED = None


def place_rank(rank, local_rank, local_world, ndev):
    """What a rank does before it allocates page-locked memory or starts threads: name its GPU and its local peers' GPUs
    (the HIP runtime's PCI names through the C ABI, include/dsp_amd.h dsp_device_pci_bdf), pin itself (pin_rank) and,
    under DSP_TIMING, say where it landed.  `local_rank` is the launcher's LOCAL_RANK (NOT reduced modulo the visible
    GPUs: ranks sharing a GPU still get different CPU slices).  Returns (bdf of my GPU or None, CPU list or None)."""
    global _PLACED
    if _PLACED is not None:   # once per process: a second call_mods in the same process must not slice its slice again
        return _PLACED
    mode = affinity_mode(local_world)
    bdf, cpus = None, None
    try:
        from . import _native
        bdf = _native.device_pci_bdf(local_rank % ndev)
        peers = local_gpu_bdfs(local_world, ndev, _native.device_pci_bdf) if mode == "numa" else None
    except Exception:   # (a library without the symbol, a runtime that cannot name the device: placement is best effort)
        peers = None
    if mode:
        cpus = pin_rank(local_rank, local_world, bdf, peers)
    # one line from rank 0 whenever the pinning CHANGED this process's affinity (it is on by default for several ranks: a
    # behaviour the user has not asked for must not be silent); every rank under DSP_TIMING
    if os.environ.get("DSP_TIMING") or (cpus is not None and rank == 0):
        import sys
        node = _numa_node_cpus(bdf)[0]
        sys.stderr.write("[dist] rank %d (local %d of %d): GPU %s, NUMA node %s, affinity %s -> CPUs %s\n" % (
            rank, local_rank, local_world, bdf, "?" if node is None else node, mode or "off",
            cpus_text(cpus) if cpus is not None else "(unchanged) " + cpus_text(sorted(os.sched_getaffinity(0)))))
    _PLACED = (bdf, cpus)
    return bdf, cpus


def cpus_text(cpus):
    """[0,1,2,3,8,9] -> "0-3,8-9" """
    if not cpus:
        return ""
    out, a, prev = [], cpus[0], cpus[0]
    for c in list(cpus[1:]) + [None]:
        if c is not None and c == prev + 1:
            prev = c
            continue
        out.append("%d" % a if a == prev else "%d-%d" % (a, prev))
        if c is not None:
            a = prev = c
    return ",".join(out)


def split_range(n, world, rank):
    """Contiguous split of n items: rank r gets [r*ceil(n/world), (r+1)*ceil(n/world)) clipped to n
    (SURVEY.md 8(e) 'Partitioning')."""
    per = (n + world - 1) // world
    a = min(n, rank * per)
    return a, min(n, a + per)


def align_to_line_start(mm, pos, size):
    """Smallest offset >= pos that begins a row (pos itself when it is 0 or follows a newline)."""
    if pos <= 0:
        return 0
    if pos >= size:
        return size
    if mm[pos - 1:pos] == b"\n":
        return pos
    nl = mm.find(b"\n", pos)
    return size if nl < 0 else nl + 1


def byte_range_for_rank(mm, size, world, rank):
    """Byte range [a, b) of a plain-text feature file owned by `rank`: the raw split points advanced to the
    next row start, so every row belongs to exactly one rank."""
    a, b = split_range(size, world, rank)
    return align_to_line_start(mm, a, size), align_to_line_start(mm, b, size)


def exclusive_prefix(counts, rank):
    return int(sum(counts[:rank]))


def comm_device(dev=None):
    """Where the tensors of a collective must live: on this rank's GPU under RCCL ("nccl"), on the host under gloo (the
    CPU tests; ranks that have to share a GPU).  This is the ONLY place the backend is looked at: every exchange of the
    path runs the same collectives on both backends (all_gather / all_reduce / ragged all_to_all_single / send + irecv),
    tensors merely hop to the host first when the group is gloo -- so that the 2- and 8-rank tests execute the production
    bookkeeping."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device(dev) if dev is not None else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def all_gather_ints(value, world, device=None):
    """One integer per rank -> list of all ranks' integers (torch.distributed must be initialised)."""
    if not collective(world):
        return [int(value)]
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(value)], dtype=torch.int64, device=comm_device(device))
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [int(x.item()) for x in out]


def all_gather_text(text, world, device=None):
    """One string per rank -> list of all ranks' strings: the lengths by one all_gather of an int, then ONE all_gather of
    uint8 tensors padded to the longest, on the collective's device (RCCL on GPU tensors, gloo on host tensors --
    comm_device).  No pickled-object collective runs on the RCCL group (until round 5 bench.py's proof-of-N-GPUs line, the
    chromosome names of the sharded call_freq and the shared-memory ring's names went through dist.all_gather_object --
    first-contact code on real hardware: pickles moved through byte tensors on whatever device the backend picks)."""
    if not collective(world):
        return [str(text)]
    import torch
    import torch.distributed as dist
    raw = str(text).encode("utf-8")
    cdev = comm_device(device)
    sizes = all_gather_ints(len(raw), world, cdev)
    width = max(max(sizes), 1)
    buf = bytearray(width)
    buf[:len(raw)] = raw
    t = torch.frombuffer(buf, dtype=torch.uint8).clone().to(cdev)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [bytes(o.cpu().numpy().tobytes())[:n].decode("utf-8") for o, n in zip(out, sizes)]


def all_gather_json(obj, world, device=None):
    """... of anything json can carry (names, small records): every rank's object, in rank order"""
    import json
    return [json.loads(t) for t in all_gather_text(json.dumps(obj), world, device)]


def _post_and_wait(ops):
    """point-to-point operations posted as ONE group (batch_isend_irecv: a single ncclGroupStart / End on RCCL, so that the
    root's receives from all ranks progress together instead of in the order they were posted), then waited for"""
    import torch.distributed as dist
    if not ops:
        return
    for q in dist.batch_isend_irecv(ops):
        q.wait()


def all_reduce_int(value, world, op="sum", device=None):
    """One integer per rank -> its sum / min / max over the ranks"""
    if not collective(world):
        return int(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(value)], dtype=torch.int64, device=comm_device(device))
    dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op])
    return int(t.item())


def all_reduce_max_float(value, world, device=None):
    if not collective(world):
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=comm_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def exchange_records(cols, dest, world, device=None):
    """Deal records -- equally long 1-D int64 columns -- to ranks: record i goes to rank dest[i].  The one real exchange
    of the call_mods path (the sharded call_freq, call_mods_freq.DeviceSiteFrequency.finish).  The columns travel as ONE
    [n, k] tensor: all_to_all_single of the per-destination counts, then ONE ragged all_to_all_single of the records
    (RCCL over xGMI on GPU tensors; gloo on host tensors -- the same two collectives).  What arrives is ordered by
    SOURCE RANK, then by the source's own order (the stable sort by destination keeps it).  Returns the columns on the
    device they came from."""
    import torch
    import torch.distributed as dist
    home = cols[0].device
    order = torch.sort(dest, stable=True)[1]
    counts = torch.bincount(dest, minlength=world)
    rec = torch.stack([c[order] for c in cols], dim=1).contiguous()
    cdev = comm_device(device if device is not None else (home if home.type == "cuda" else None))
    counts, rec = counts.to(cdev), rec.to(cdev)
    recv_counts = torch.empty_like(counts)
    dist.all_to_all_single(recv_counts, counts)
    ins, outs = counts.tolist(), recv_counts.tolist()
    got = torch.empty((sum(outs), len(cols)), dtype=rec.dtype, device=cdev)
    dist.all_to_all_single(got, rec, outs, ins)
    got = got.to(home)
    return [got[:, j].contiguous() for j in range(len(cols))]


def gather_probs(probs, world, dst=0):
    """Optional final gather of per-site probabilities [n_r, C] to rank `dst` (ragged: one integer per rank is exchanged
    first).  A TRUE gather: every other rank sends its own rows to `dst` point to point (RCCL send/recv over xGMI on
    GPU tensors, gloo on host tensors -- comm_device), `dst` receives each rank's rows into a tensor of exactly that
    rank's size.  Off-root memory stays O(n_r) -- round 3's padded all_gather made every rank hold world x max(n_r) rows
    that only `dst` read (800 MB per rank at BASELINE configs[3]).  Returns the list of the ranks' tensors on `dst`
    (on the collective's device), None elsewhere; ~8 B/site."""
    if not collective(world):
        return [probs]
    import torch
    import torch.distributed as dist
    cdev = comm_device(probs.device if probs.is_cuda else None)
    sizes = all_gather_ints(probs.shape[0], world, cdev)
    me = dist.get_rank()
    probs = probs.contiguous().to(cdev)
    if me != dst:
        if sizes[me]:
            _post_and_wait([dist.P2POp(dist.isend, probs, dst)])
        return None
    out = [probs if r == me else torch.empty((sizes[r],) + tuple(probs.shape[1:]), dtype=probs.dtype, device=cdev)
           for r in range(world)]
    _post_and_wait([dist.P2POp(dist.irecv, out[r], r) for r in range(world) if r != me and sizes[r]])
    return out


def gather_columns(cols, world, device=None):
    """Ragged gather of equally long 1-D int64 columns to rank 0: sizes exchanged first, then every other rank sends its
    columns, stacked as [n, k], point to point (RCCL on GPU tensors, gloo on host tensors -- comm_device) -- a true gather
    like gather_probs: off-root memory stays O(n_r) (until round 4 a padded all_gather: world x max(n_r) rows on every
    rank, read by rank 0 only).  Returns the concatenated columns on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    n = int(cols[0].numel())
    cdev = comm_device(device if device is not None else (cols[0].device if cols[0].is_cuda else None))
    sizes = all_gather_ints(n, world, cdev)
    rec = torch.stack([c.to(cdev) for c in cols], dim=1).contiguous()
    me = dist.get_rank()
    if me != 0:
        if n:
            _post_and_wait([dist.P2POp(dist.isend, rec, 0)])
        return None
    got = [rec if r == 0 else torch.empty((sizes[r], len(cols)), dtype=rec.dtype, device=cdev) for r in range(world)]
    _post_and_wait([dist.P2POp(dist.irecv, got[r], r) for r in range(1, world) if sizes[r]])
    out = torch.cat(got)
    return [out[:, j].contiguous() for j in range(len(cols))]
