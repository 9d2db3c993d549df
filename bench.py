#!/usr/bin/env python3
"""bench.py -- methylation sites/s of the call_mods forward (both_bilstm bn13_sn16) on N x MI355X.

A "step" is one pass of the hot path over one batch of 65,536 synthetic sites whose feature tensors are
already resident in HBM (BASELINE.json configs[1]: 10M synthetic sites, fp32, batch 65536; 153 steps =
10,027,008 sites).  Sites are independent, so N GPUs range-shard the site index space with no data-path
collective ("weak" scaling: every rank runs K steps of its own range).  One JSON line on rank 0.

`python bench.py --gpus N` started plainly starts its N ranks itself: a fresh `torch.distributed.run` child
(one process per GPU, RCCL) created BEFORE this process makes any GPU call -- the same pattern as
`call_mods --nproc_gpu N` (deepsignal_plant_amd/call_modifications.py:_self_launch; the reference starts its own
model processes, call_modifications.py:613-621).  Started under a launcher (WORLD_SIZE set) it is one rank.

Config 3 of BASELINE.json (seq-only branch, hid 256 x 2 layers): `--model_type seq_bilstm --layernum1 2`.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 65536
FP32_MATRIX_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
CPU_BASELINE_TARGET_S = float(os.environ.get("DSP_CPU_BASELINE_S", 25.0))  # bounded sample: at most ~25 s of host work
# The reference itself cannot travel to the GPU box; its own numbers were measured in the survey container
# (8 vCPU Xeon 2.1 GHz, torch CPU) and are carried next to the port's, labelled.
REFERENCE_CPU = {"hardware": "8 vCPU Xeon 2.10 GHz (survey container, BASELINE.md section 2), not this box",
                 "call_mods_default_flags_sites_per_s": 237.0,
                 "call_mods_omp4_x2procs_sites_per_s": 1640.0,
                 "forward_only_b512_8threads_sites_per_s": 2401.0,
                 "config": "both_bilstm bn13_sn16, 100k-row feature TSV, batch 512 (BASELINE.json configs[0])"}
KERNEL_SOURCES = ("deepsignal_plant_amd/csrc/dsp_kernels.hip", "deepsignal_plant_amd/csrc/dsp_kernels.h",
                  "deepsignal_plant_amd/csrc/dsp_cluster_protocol.h", "deepsignal_plant_amd/csrc/dsp_capi.cpp")


def kernel_source_hash():
    """sha256 over the CODE of the sources that determine the forward's kernels and launch geometry (// comments and
    blank lines are dropped first, so that editing a comment does not orphan a measurement): a committed PMC traffic
    figure is only quoted while it was measured on exactly this code."""
    import re
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "r") as f:
            for line in f:
                line = re.sub(r"//.*$", "", line).strip()
                if line:
                    h.update(line.encode())
                    h.update(b"\n")
    return h.hexdigest()[:16]


def committed_traffic(args, stale=False):
    """The committed rocprofv3 PMC entry of this same command (profiles/traffic.json, written by tools/make_traffic.py on
    the GPU box): HBM-side bytes per launch of the dominant kernel and per step over all kernels.  None unless the entry
    was measured on the current kernel sources and on this workload.  stale=True: the entry of this workload measured on
    OTHER sources (reported apart, under its own hash, never as `traffic`)."""
    try:
        entries = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        return None
    want = {"model_type": args.model_type, "layernum1": args.layernum1, "hid_rnn": args.hid_rnn, "batch": args.batch,
            "precision": args.precision}
    now = kernel_source_hash()
    for e in entries:
        if all(e.get(k) == v for k, v in want.items()) and (e.get("kernel_src_sha16") == now) != stale:
            return e
    return None


def cpu_baseline(model_cfg_kwargs, sd_numpy, seed):
    """BASELINE.json configs[0]'s shape on this box's host cores: a 100,000-row synthetic feature
    TSV -> row parser -> forward in 512-row chunks (the reference's --batch_size, call_modifications.py:147) -> per-read
    call lines -> file.  The forward is the oracle's C port (oracle/dsp_oracle.c, fp32, OpenMP over site blocks -- kind
    "port"); parsing and formatting are this build's host code (csrc/dsp_text.cpp), the same the GPU pipeline uses.
    Bounded: rows are processed in file order until the time budget is spent (DSP_CPU_BASELINE_S, default 25 s); the
    sample string says how many of the 100,000 were."""
    import tempfile

    import numpy as np

    from deepsignal_plant_amd import textio
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    cfg = onp.OracleConfig(**model_cfg_kwargs)
    threads = oc.num_threads()
    n_rows, chunk = 100000, 512
    tmp = tempfile.mkdtemp(prefix="dsp_cpu_baseline_")
    path = os.path.join(tmp, "features_100k.tsv")
    try:
        # the input file is made outside the timed region (tools/make_tsv.py: one writer process per core)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_tsv.py"), path, str(n_rows), "--seed", "5"],
                              stdout=subprocess.DEVNULL)   # (bench.py prints ONE line)
        size = os.path.getsize(path)
        t0 = time.time()
        data = np.memmap(path, dtype=np.uint8, mode="r")   # parsed in place from the page cache, as the product's reader does
        rows = textio.parse_rows(data, cfg.seq_len, cfg.signal_len, nthreads=threads)
        t_parse = time.time() - t0
        assert rows.n == n_rows
        probs = np.empty((n_rows, cfg.num_classes), np.float32)
        done = 0
        t1 = time.time()
        while done < n_rows and (done == 0 or (time.time() - t0) < 0.6 * CPU_BASELINE_TARGET_S):
            a, b = done, min(n_rows, done + chunk)
            _lg, pr = oc.forward(cfg, sd_numpy, rows.kmer[a:b].astype(np.float32), rows.means[a:b], rows.stds[a:b],
                                 rows.lens[a:b].astype(np.float32), rows.signals[a:b], init_mode="philox", seed=seed,
                                 site_offset=a, nthreads=threads)
            probs[a:b] = pr
            done = b
        t_fwd = time.time() - t1
        t2 = time.time()
        labels = probs[:done].argmax(1).astype(np.uint8)
        text = textio.format_calls(rows, probs[:done], labels, nthreads=threads, start=0, stop=done)
        with open(os.path.join(tmp, "calls.tsv"), "wb") as f:
            f.write(text)
        t_fmt = time.time() - t2
        # the parse covered the whole file: charge the sample its share of it
        dt = t_parse * done / n_rows + t_fwd + t_fmt
        # second figure, labelled: the same port fed LARGE batches -- the oracle parallelises over blocks of sites,
        # so a 512-row chunk leaves most cores of a big host idle; this is what the host could do if the reference's batch
        # size were not 512 (forward only, rounds 1-3's definition of the baseline)
        big, big_done, tb = 8192, 0, time.time()
        while big_done + big <= n_rows and (big_done == 0 or (time.time() - tb) < 0.4 * CPU_BASELINE_TARGET_S):
            a, b = big_done, big_done + big
            oc.forward(cfg, sd_numpy, rows.kmer[a:b].astype(np.float32), rows.means[a:b], rows.stds[a:b],
                       rows.lens[a:b].astype(np.float32), rows.signals[a:b], init_mode="philox", seed=seed, site_offset=a,
                       nthreads=threads)
            big_done = b
        t_big = time.time() - tb
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return {"value": round(done / dt, 1), "unit": "sites/s", "cores": threads, "cpu_model": _cpu_model(), "kind": "port",
            "sample": "the first %d of a 100,000-row synthetic feature TSV (%.0f MB; BASELINE.json configs[0]'s shape): parse "
                      "%.2f s for the file + forward in %d-row chunks %.2f s (oracle/dsp_oracle.c fp32 + OpenMP, same "
                      "model/weights, Philox N(0,1) states) + format and write %.2f s" % (done, size / 1e6, t_parse, chunk, t_fwd, t_fmt),
            "seconds": {"parse_whole_file": round(t_parse, 3), "forward": round(t_fwd, 3), "format_write": round(t_fmt, 3)},
            "chunk_rows": chunk,
            "port_forward_only_large_batches": {"value": round(big_done / t_big, 1) if big_done else None, "unit": "sites/s",
                                                "chunk_rows": big, "sites": big_done, "seconds": round(t_big, 3),
                                                "what": "the same C port, forward only, %d-row batches (every core busy): "
                                                        "not the reference's configuration, which feeds 512" % big},
            "reference_proper": REFERENCE_CPU}


def _rccl_version(torch):
    """version of the RCCL library torch is linked against ("nccl" IS RCCL on ROCm), e.g. "2.26.6" """
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:
        return "unknown (%r)" % (e,)


def _cpu_model():
    """the host CPU's name, printed next to the CPU baseline"""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=153)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--model_type", default="both_bilstm", choices=["both_bilstm", "seq_bilstm", "signal_bilstm"])
    ap.add_argument("--layernum1", type=int, default=3)
    ap.add_argument("--layernum2", type=int, default=1)
    ap.add_argument("--hid_rnn", type=int, default=256)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--precision", default=os.environ.get("DSP_PRECISION", "fp32"), choices=["fp32", "bf16x6", "bf16x9", "fp16x3"],
                    help="products of the combined stack: fp32 MFMA (default, what `value` is measured in) or the opt-in "
                         "split-bf16 emulation (include/dsp_amd.h DSP_PREC_*)")
    ap.add_argument("--no_alt", action="store_true", help="skip the extra split-precision measurement reported under alt_precision")
    ap.add_argument("--n1_ms", type=float, default=None,
                    help="ms_per_step of the N=1 run of this same command: the line then carries scaling_efficiency_vs_n1 = "
                         "n1_ms / ms_per_step (weak scaling: every rank does the N=1 run's work)")
    ap.add_argument("--gather", action="store_true",
                    help="optional final gather of EVERY step's per-site probabilities to rank 0 (dist.gather_probs: RCCL "
                         "send/recv over xGMI, off-root memory O(own rows)); outside the timed region, reported under `gather`")
    return ap.parse_args(argv)


def self_launch(args, argv):
    """--gpus N without a launcher: start N ranks of this script as a fresh child (this process has made no GPU call,
    and never will) and return the child's exit code.  None = this process is a rank (or N == 1)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return None
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    rc = self_launch(args, argv)
    if rc is not None:
        return rc

    import torch
    import torch.distributed as dist
    from deepsignal_plant_amd import dist as dsp_dist
    from deepsignal_plant_amd import synth
    from deepsignal_plant_amd.models import ModelBiLSTM

    rank, local_rank, world = dsp_dist.env_world()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start it plainly, or under torch.distributed.run "
                         "with --nproc-per-node equal to --gpus)" % (args.gpus, world))
    ndev = torch.cuda.device_count()
    assert ndev > 0, "bench.py needs an MI355X"
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    my_bdf, my_cpus = dsp_dist.place_rank(rank, local_rank, local_world, ndev)   # NUMA placement by default when world > 1
    dev_index = local_rank % ndev
    # one process per GPU over RCCL; if fewer GPUs than ranks are visible (the 1-GPU dev box), the ranks share GPUs and
    # the control-plane collectives (barrier, max of the wall time) run over gloo instead.  DSP_FORCE_DIST=1 makes a lone
    # rank build its own one-rank RCCL group and take the same collective branches (deepsignal_plant_amd/dist.py)
    backend = dsp_dist.init_process_group(world, rank, dev_index, ndev)
    multi = dsp_dist.collective(world)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    K, W, B = args.steps, args.warmup, args.batch
    model = ModelBiLSTM(13, 16, args.layernum1, args.layernum2, 2, 0, args.hid_rnn, 16, 4, True, True, module=args.model_type,
                        device=dev_index, init_state="randn", seed=2024)
    sd = synth.random_state_dict(model, seed=1234)
    model.load_state_dict(sd)
    model.cuda(dev_index).eval()
    model.set_precision(args.precision)
    model.reserve(B)
    flops_site = model.flops_per_site()

    # resident synthetic input: distinct batches (up to 160 = 10.9 GB); more steps cycle over them
    nb = min(K, 160) if K > 0 else 1
    batches = [synth.feature_batch(B, device=str(dev), seed=1000 * rank + i) for i in range(nb)]
    site0, site1 = dsp_dist.split_range(world * K * B, world, rank)  # this rank's range of the global site index space
    outs = None
    # --gather: every step's probabilities are kept (8 B per site, a device-to-device copy of 0.5 MB per step, on the
    # compute stream) so that the final gather moves the whole run's calls, as BASELINE configs[3] describes it
    kept = torch.empty((K * B, model.num_classes), dtype=torch.float32, device=dev) if args.gather and K > 0 else None

    def step(i, keep=False):
        model.site_offset = site0 + i * B
        o = model(*batches[i % nb])
        if keep and kept is not None:
            kept[i * B:(i + 1) * B].copy_(o[1], non_blocking=True)
        return o

    for i in range(W):
        step(i)
    torch.cuda.synchronize()
    # HIP events on the launch stream around the run of the DOMINANT kernel's launches only (the roofline object's duration is
    # measured live, over the timed region): ONE pair of records per forward, its time shared equally by the run's launches (the
    # gaps and, on small batches, the clean-up launches between them included: conservative).  The per-launch breakdown of ALL
    # launches is taken in the untimed steps below -- until round 5 it rode on the timed steps too: 9 records, 0.03 ms per
    # forward whatever the batch.
    model.profile("dominant")
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        outs = step(i, True)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = model.profile_read()
    model.profile(False)
    # the same steps once more WITHOUT any events (untimed by the contract; reported beside the line): an event record costs the
    # stream ~4 us -- nothing at 65,536 sites, 1-2 % at 512 with the 2 records the timed steps keep
    ms_events_off = None
    prof_all = []
    if world == 1 and K > 0:
        ko = min(K, 40)
        for i in range(min(W, 5) + 3):      # (reading the events back left the GPU idle: back to its clocks first)
            step(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(ko):
            step(i)
        torch.cuda.synchronize()
        ms_events_off = (time.perf_counter() - t1) / ko * 1e3
        # ... and once more with events around EVERY launch: the per-launch breakdown (ms_per_step_by_launch)
        model.profile(True)
        for i in range(ko):
            step(i)
        torch.cuda.synchronize()
        prof_all = model.profile_read()
        model.profile(False)

    ranges = [[site0, site1]]
    gather_info = None
    devices = None
    if multi:
        # every collective below is the same on RCCL and on gloo (dist.comm_device: tensors hop to the host under gloo)
        dt_mine = dt
        dt = dsp_dist.all_reduce_max_float(dt, world, dev)
        a = dsp_dist.all_gather_ints(site0, world, dev)
        b = dsp_dist.all_gather_ints(site1, world, dev)
        ranges = [[x, y] for x, y in zip(a, b)]
        # proof of N GPUs: every rank names the device it ran on (PCI name and uuid from the HIP runtime, through the C ABI),
        # its host, its CPUs and its OWN wall time; with RCCL the PCI names must be pairwise distinct per host
        from deepsignal_plant_amd import _native
        try:
            my_uuid = _native.device_uuid(dev_index)
        except Exception as e:   # (identity is evidence, not a reason to lose the measurement)
            my_uuid = "unknown (%s)" % type(e).__name__
        ident = {"rank": rank, "host": socket.gethostname(), "local_rank": local_rank, "hip_device": dev_index,
                 "pci_bdf": my_bdf, "uuid": my_uuid, "name": torch.cuda.get_device_name(dev_index),
                 "numa_node": dsp_dist._numa_node_cpus(my_bdf)[0],
                 "cpus": dsp_dist.cpus_text(my_cpus if my_cpus is not None else sorted(os.sched_getaffinity(0))),
                 "pinned": my_cpus is not None, "ms_per_step": round(dt_mine / max(K, 1) * 1e3, 3),
                 "visible_devices": ndev}
        # (one fixed-width byte tensor per rank through dist.comm_device: no pickled-object collective on the RCCL group)
        devices = dsp_dist.all_gather_json(ident, world, dev)
        # a device the runtime could not name counts by its index on its host (the check must not refuse a good run)
        key = lambda d: (d["host"], d["pci_bdf"], d["uuid"]) if d["pci_bdf"] else (d["host"], "hip device %d" % d["hip_device"])
        distinct = len({key(d) for d in devices}) == world
        must = backend == "nccl" or os.environ.get("DSP_REQUIRE_DISTINCT_GPUS") == "1"
        if must and world > 1 and not distinct:
            if rank == 0:
                sys.stderr.write("bench.py: %d ranks but only %d distinct GPUs: %s -- a SCALE line from this run would not "
                                 "measure %d GPUs; refusing to print one\n" % (
                                     world, len({key(d) for d in devices}),
                                     [(d["rank"], d["pci_bdf"]) for d in devices], world))
            dist.barrier()
            dist.destroy_process_group()
            return 3
        if kept is not None:  # the optional final gather of the run's per-site probabilities (a true gather to rank 0)
            torch.cuda.synchronize()
            dist.barrier()
            tg = time.perf_counter()
            got = dsp_dist.gather_probs(kept, world)   # (tensors on this GPU over RCCL, on the host over gloo: dist.comm_device)
            torch.cuda.synchronize()
            tg = time.perf_counter() - tg
            if rank == 0:
                rows = [int(g.shape[0]) for g in got]
                assert rows == [y - x for x, y in ranges], (rows, ranges)   # rank r's rows ARE its site range
                ok = all(bool(torch.isfinite(g).all()) and bool(((g.sum(1) - 1).abs() < 1e-5).all()) for g in got)
                assert ok and torch.equal(got[0].to(kept.device), kept)
                gather_info = {"sites": sum(rows), "bytes": sum(rows) * 8, "seconds": round(tg, 4), "to_rank": 0,
                               "how": "dist.gather_probs: sizes by one all_gather of an int, then send/recv of each rank's own rows"}
    assert K == 0 or (outs is not None and bool(torch.isfinite(outs[1]).all()))

    # opt-in mode, reported next to the headline (never as `value`): the same steps with the LSTMs' fp32 products
    # emulated by split low-precision pieces on the fast matrix pipes (include/dsp_amd.h DSP_PREC_*), and how far
    # its probabilities are from the fp32 path's
    alt = None
    if world == 1 and args.precision == "fp32" and not args.no_alt and K > 0:
        alt = []
        model.site_offset = site0
        ref = model(*batches[0])[1].clone()
        for mode, what in (("bf16x9", "3 bf16 pieces per operand, all 9 piece products (exact products), f32 accumulate"),
                           ("fp16x3", "2 fp16 pieces per operand, 3 piece products, f32 accumulate; front ends bf16x6")):
            ka = min(K, 40)
            try:
                model.set_precision(mode)
            except Exception as e:  # e.g. fp16x3 refused for this checkpoint
                alt.append({"dtype": "f32 via %s" % mode, "value": None, "why": repr(e)})
                continue
            for i in range(2):
                step(i)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for i in range(ka):
                step(i)
            torch.cuda.synchronize()
            ta = time.perf_counter() - ta
            model.site_offset = site0
            dmax = float((model(*batches[0])[1] - ref).abs().max())
            alt.append({"dtype": "f32 via %s (%s)" % (mode, what), "value": round(ka * B / ta, 1), "unit": "sites/s",
                        "steps": ka, "ms_per_step": round(ta / ka * 1e3, 3), "max_abs_dprob_vs_fp32_path": dmax,
                        "how": "python bench.py --precision %s" % mode})
        model.set_precision("fp32")

    if rank == 0:
        total_sites = world * K * B
        value = total_sites / dt if dt > 0 else 0.0
        # roofline of the dominant kernel = the launches of the combined BiLSTM stack (the front-end launches run a
        # different instantiation and are listed separately by rocprofv3): algorithmic FLOPs of those launches /
        # their summed duration, durations from HIP events on the launch stream
        comb_ms = [ms for name, ms in prof if name == "lstm_comb"]
        k_all = min(K, 40) if prof_all else 0
        all_ms = sum(ms for _, ms in prof_all) / max(k_all, 1) * K if prof_all else sum(ms for _, ms in prof)
        H, T = model.hidden_size, model.seq_len
        comb_flops_site = sum(2 * (2 * T * 4 * H * ((H if k == 0 else 2 * H) + H)) for k in range(model.num_layers1))
        n_lstm = max(len(comb_ms), 1)
        avg_ms = sum(comb_ms) / n_lstm
        flops_per_launch = comb_flops_site * B * K / n_lstm
        nprod = {"fp32": 1, "bf16x6": 6, "bf16x9": 9, "fp16x3": 3}[args.precision]
        # split-bf16 modes execute nprod bf16 piece products per fp32 product: price those against the bf16 peak
        peak = FP32_MATRIX_PEAK_TFLOPS if nprod == 1 else 16 * FP32_MATRIX_PEAK_TFLOPS
        achieved = nprod * flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        per_launch = {}
        for name, ms in (prof_all or prof):
            per_launch.setdefault(name, []).append(ms)
        # HBM-side traffic from the committed PMC passes of this command, next to the bytes the algorithm needs: per launch
        # of the dominant kernel its K4 activations (x read by both directions' workgroups + h written), per step SURVEY.md
        # 8(d)'s 1,048 B per site (features in, probabilities out)
        tr = committed_traffic(args)
        traffic = tr.get("hbm_bytes_per_launch") if tr else None
        step_traffic = tr.get("hbm_bytes_per_step_all_kernels") if tr else None
        old = None if tr else committed_traffic(args, stale=True)
        k4_site = sum(2 * (H if k == 0 else 2 * H) * T * 4 + 2 * H * T * 4 for k in range(model.num_layers1)) / max(model.num_layers1, 1)
        algo_launch = k4_site * B
        cfg_idx = 2 if args.model_type == "seq_bilstm" else (1 if world == 1 else 3)
        line = {
            "metric": "methylation sites/sec, %s bn13_sn16" % args.model_type, "value": round(value, 1), "unit": "sites/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / max(K, 1) * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if nprod == 1 else "f32 via %s (%s pieces per operand, %d piece products, f32 accumulate)" % (
                args.precision, "2 fp16" if nprod == 3 else "3 bf16", nprod),
            "data": "synthetic",
            "config": {"workload": "%d synthetic sites, %s bn13_sn16 fp32, batch %d on %dxMI355X (BASELINE.json configs[%d])"
                                   % (total_sites, args.model_type, B, world, cfg_idx),
                       "batch": B, "sites": total_sites, "layernum1": args.layernum1, "layernum2": args.layernum2,
                       "hid_rnn": args.hid_rnn,
                       "init_state": "in-kernel Philox N(0,1) (stand-in for torch.randn, models.py:169-176)",
                       "weights": "seeded random state_dict, PyTorch default-init scale", "parallelism": "range-shard x%d" % world,
                       "backend": ("rccl" if backend == "nccl" else "gloo (ranks share %d GPU)" % ndev) if multi else "none",
                       "rank_site_ranges": ranges, "flops_per_site": flops_site,
                       **({"rccl_version": _rccl_version(torch) if backend == "nccl" else None,
                           "distinct_gpus": len({((d["host"], d["pci_bdf"], d["uuid"]) if d["pci_bdf"] else (d["host"], d["hip_device"])) for d in devices}),
                           "ranks": devices} if devices else {})},
            "roofline": {"bound": "mfma", "kernel": "dsp_lstm_kernel (combined stack)" if nprod == 1 else "dsp_lstm_split_kernel<%d>" % nprod,
                         "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "traffic": traffic, "algorithmic_bytes_per_launch": algo_launch,
                         "traffic_over_algorithmic": round(traffic / algo_launch, 3) if traffic else None,
                         "traffic_per_step_all_kernels": step_traffic, "algorithmic_bytes_per_step": 1048 * B,
                         "step_traffic_over_algorithmic": round(step_traffic / (1048 * B), 1) if step_traffic else None,
                         "avg_launch_ms": round(avg_ms, 4), "launches": len(comb_ms),
                         "flops_per_launch": flops_per_launch,
                         "whole_forward_tflops": round(value / world * flops_site / 1e12, 2),
                         "whole_forward_frac": round(value / world * flops_site / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4),
                         "hbm_gbps_algorithmic": round(value / world * 1048 / 1e9, 3),
                         "hbm_frac_algorithmic": round(value / world * 1048 / 8e12, 6),  # of 8 TB/s: the north_star's "HBM roofline" does not bind
                         "kernel_time_frac_of_wall": round(all_ms * 1e-3 / dt, 4) if (dt > 0 and prof_all) else None,
                         "ms_per_step_events_off": round(ms_events_off, 4) if ms_events_off else None,
                         "whole_forward_frac_events_off": round(B / (ms_events_off * 1e-3) * flops_site / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4)
                         if ms_events_off else None,
                         "ms_per_step_by_launch": {k: round(sum(v) / max(k_all or K, 1), 4) for k, v in per_launch.items()},
                         "events_in_timed_steps": "one pair per forward around the dominant kernel's launches; ms_per_step_by_launch from %s" % (
                             "%d untimed steps with events around every launch" % k_all if k_all else "the timed steps"),
                         "kernel_src_sha16": kernel_source_hash(),
                         **({"traffic_measured_on_earlier_sources": {
                             "hbm_bytes_per_launch": old.get("hbm_bytes_per_launch"), "hbm_bytes_per_step_all_kernels": old.get("hbm_bytes_per_step_all_kernels"),
                             "kernel_src_sha16": old.get("kernel_src_sha16"),
                             "note": "NOT this build's traffic (hence traffic: null): the PMC passes of this workload on the sources of the hash given; "
                                     "re-measure with tools/profile.sh + tools/make_traffic.py"}} if old else {}),
                         "note": ("fp32 MFMA and VALU work do not overlap on gfx950 (profiles/r2/micro_mfma_cell_overlap.txt): "
                                 "with the LSTM cell phase counted the bound of this kernel is 0.974 of the MFMA peak "
                                 "(DESIGN.md section 3)") if nprod == 1 else None},
        }
        if args.n1_ms:
            line["scaling_efficiency_vs_n1"] = round(args.n1_ms / (dt / max(K, 1) * 1e3), 4)
            line["n1_ms_per_step"] = args.n1_ms
        if gather_info:
            line["gather"] = gather_info
        if alt:
            line["alt_precision"] = alt
        if not args.no_cpu_baseline and world == 1:
            try:
                kw = dict(num_layers1=args.layernum1, num_layers2=args.layernum2, hidden_size=args.hid_rnn, module=args.model_type)
                line["cpu_baseline"] = cpu_baseline(kw, {k: v.numpy() for k, v in sd.items()}, 2024)
            except Exception as e:  # the baseline must never take the GPU number down with it
                line["cpu_baseline"] = {"value": None, "unit": "sites/s", "cores": 0, "kind": "port",
                                        "sample": "failed: %r" % (e,), "reference_proper": REFERENCE_CPU}
        print(json.dumps(line), flush=True)
    if multi:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
