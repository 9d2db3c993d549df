"""call modifications from extracted features (feature TSV in -> per-read-call TSV out) on MI355X.

Host-side mirror of deepsignal_plant/call_modifications.py for the TSV branch of ``call_mods``
(:584-636): same entry point ``call_mods(args)``, same flags, same input row grammar (:76-86) and the same
output grammar (chromosome, pos, strand, pos_in_strand, read_name, read_strand, prob_0, prob_1,
called_label, 5-mer; :175-188, :262-282).  What changed underneath:

  reference                                         this build
  ------------------------------------------------  ---------------------------------------------------------
  reader process, per-row Python, pickled lists     FeatureReader thread + native multi-threaded parser into
  over mp.Queue (:55-127)                            pinned SoA buffers (feed.py, csrc/dsp_text.cpp)
  N model processes x (5 sync H2D copies +          one process per GPU: async H2D on the compute stream's
  nn.LSTM forward + 1 sync D2H) per 512 rows         predecessor, one fused-kernel forward per block through
  (:130-170, :195-259)                               the C ABI, async D2H of probs/labels (models.py)
  per-row Python string building (:175-188)         native formatter, byte-identical strings
  writer process (:262-282)                          writer thread, rows in INPUT order

Multi-GPU: contiguous byte-range split of the file over ranks (dist.py); per-rank part files concatenated by
rank 0.  Results do not depend on the number of GPUs or on batching: the in-kernel initial states are keyed
by the global row index (feature files) or by (read uid, base index in the read) (the fast5-directory branch, :559-583,
which extracts features on the GPU from read records: _call_mods_reads).
"""
from __future__ import annotations

import argparse
import os
import queue
import sys
import threading
import time

import numpy as np

from . import dist as dsp_dist
from . import canary, featfile, feed, gzio, textio
from .models import ModelBiLSTM
from .utils.process_utils import display_args, str2bool


def _writer_delay():
    """DSP_WRITER_DELAY_MS (tests): a writer that cannot keep up with the GPU"""
    return float(os.environ.get("DSP_WRITER_DELAY_MS", "0") or 0) / 1e3
_TICKS = []   # DSP_TIMING=1: (label, seconds since call_mods started) of the run's milestones, printed by rank 0 at the end


def _tick(label):
    if os.environ.get("DSP_TIMING"):
        _TICKS.append((label, time.time()))


def _get_gpus():
    """Visible GPUs (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES honoured); the reference's round-robin list at
    call_modifications.py:523-529.  Counted without loading the HIP runtime (dist.visible_gpu_count)."""
    return list(range(dsp_dist.visible_gpu_count()))


_STATE_KEYS = ("h_seq", "c_seq", "h_sig", "c_sig", "h_comb", "c_comb")


def init_state_mode(args):
    """--init_state: ("randn" | "zeros", None) or ("file", path) for file:<states.npz>"""
    v = getattr(args, "init_state", "randn") or "randn"
    if v in ("randn", "zeros"):
        return v, None
    if v.startswith("file:") and len(v) > 5:
        return "file", v[5:]
    raise ValueError("--init_state must be randn, zeros or file:<states.npz>")


class FileInitStates(object):
    """--init_state file:<npz>: the initial LSTM states of every input row, in the layout init_hidden returns them
    (models.py:169-176): h_seq / c_seq (2*layernum2, N, hid_seq), h_sig / c_sig (2*layernum2, N, hid_signal), h_comb /
    c_comb (2*layernum1, N, hid_rnn), row n of axis 1 = the n-th row of the input.  Lets a captured reference run -- the
    draws of torch.randn under a known seed -- be replayed end to end through `call_mods` (the keys may carry the
    `state_` prefix of the golden fixtures, tests/golden/f1_*.npz).  16 KiB per row: a replay facility, not a run mode."""

    def __init__(self, path, model):
        if not os.path.exists(path):
            raise ValueError("--init_state file: %s does not exist!" % path)
        z = np.load(path, allow_pickle=False)
        H = model.hidden_size
        hs, hg = (H // 2, H - H // 2) if model.module == "both_bilstm" else ((H, 0) if model.module == "seq_bilstm" else (0, H))
        want = {"h_seq": (2 * model.num_layers2, hs), "c_seq": (2 * model.num_layers2, hs), "h_sig": (2 * model.num_layers2, hg),
                "c_sig": (2 * model.num_layers2, hg), "h_comb": (2 * model.num_layers1, H), "c_comb": (2 * model.num_layers1, H)}
        self.arrays, self.rows = {}, None
        for k, (nl, hid) in want.items():
            if hid == 0:
                continue
            name = k if k in z.files else "state_" + k
            if name not in z.files:
                raise ValueError("--init_state file: %s holds no array %s" % (path, k))
            a = np.asarray(z[name], np.float32)
            if a.ndim != 3 or a.shape[0] != nl or a.shape[2] != hid:
                raise ValueError("--init_state file: %s is %s, the model needs (%d, rows, %d)" % (k, a.shape, nl, hid))
            if self.rows is not None and a.shape[1] != self.rows:
                raise ValueError("--init_state file: %s holds %d rows, the arrays before it %d" % (k, a.shape[1], self.rows))
            self.rows = int(a.shape[1])
            self.arrays[k] = a
        self.path = path

    def for_rows(self, first_row, n):
        import torch
        if first_row + n > self.rows:
            raise ValueError("--init_state file: %s holds the states of %d rows, the input has more" % (self.path, self.rows))
        return {k: torch.from_numpy(np.ascontiguousarray(a[:, first_row:first_row + n])) for k, a in self.arrays.items()}


def parse_on(args, nthreads=None):
    """--parse_on: where feature rows become numbers.  "device" (default; DSP_PARSE_ON overrides the default): one GPU thread
    per token parses the staged text (csrc/dsp_parse_dev.hip: 0.17 ms of GPU time per 32,768 rows = 0.65 % of their forward,
    0.4-0.7 us of host time per row -- one host thread per GPU); "host": this rank's parser threads (2.5-3.6 us of host time
    per row: four threads keep up with one GPU).  Same values either way; end to end the two are equally fast on one GPU
    (4 M rows: 3.73 s device with one host thread, 3.71 s host with four)."""
    v = getattr(args, "parse_on", None) or os.environ.get("DSP_PARSE_ON") or "device"
    if v == "auto":
        v = "device"
    if v not in ("device", "host"):
        raise ValueError("--parse_on must be device or host")
    return v


def load_model(args, device):
    """Replaces the head of _call_mods_q (call_modifications.py:214-228)."""
    import torch
    mode, _ = init_state_mode(args)
    model = ModelBiLSTM(args.seq_len, args.signal_len, args.layernum1, args.layernum2, args.class_num,
                        args.dropout_rate, args.hid_rnn, args.n_vocab, args.n_embed, str2bool(args.is_base),
                        str2bool(args.is_signallen), module=args.model_type, device=device,
                        init_state="randn" if mode == "file" else mode, seed=getattr(args, "seed", 0))
    para_dict = torch.load(args.model_path, map_location=torch.device('cpu'))
    model_dict = model.state_dict()
    model_dict.update(para_dict)
    model.load_state_dict(model_dict)
    model.cuda(device)
    model.eval()
    if getattr(args, "precision", None):
        model.set_precision(args.precision)
    return model


class _Writer(threading.Thread):
    """Replaces _write_predstr_to_file (:262-282): formats finished blocks natively and appends them in order."""

    def __init__(self, path, is_gzip, nthreads, reader, freq=None):
        super().__init__(daemon=True)
        self.q = queue.Queue(maxsize=4)
        self.path, self.is_gzip, self.nthreads, self.reader = path, is_gzip, nthreads, reader
        self.freq = freq  # optional SiteFrequency (host) fed straight from the GPU results (fused call_freq)
        self.freq_dev = None  # optional (DeviceSiteFrequency, stream, first-row getter): the reads branch feeds it from here
        self.error = None
        self.rows = 0
        self.delay = _writer_delay()
        self.canary = canary.on()  # DSP_SLOT_CANARY=1 (canary.py): what the writer reads must not be a slot already given back
        self.class_num = 2
        self.mark_blocks = False   # remember where every block's output ends in the part file (interleaved sharding)
        self.n_vocab = 16          # --n_vocab: rows whose k-mer holds a code beyond it are an error (set by the caller)
        self.block_ends = []

    def _check_live(self, block, probs, labels):
        """DSP_SLOT_CANARY: the block's input slot and result slot must still be this block's (no poison in what is read)"""
        rows, n = block.rows, block.rows.n
        named = [("probs (result slot)", probs), ("labels (result slot)", labels), ("kmer", np.asarray(rows.kmer)[:n]),
                 ("info_len", np.asarray(rows.info_len)[:n]), ("row_off", np.asarray(rows.row_off)[:n])]
        if isinstance(rows.text, np.ndarray) and n:
            a, b = int(rows.row_off[0]), int(rows.row_off[n - 1]) + int(rows.info_len[n - 1])
            named.append(("sampleinfo text", rows.text[a:b]))
        canary.expect_live(named, "writer formats block of first row %d" % block.first_row)
        if int(labels.max()) >= self.class_num:
            raise canary.SlotCanaryError("DSP_SLOT_CANARY: a label of block %d is no class" % block.first_row)

    def run(self):
        try:
            wf = gzio.open_write(self.path, self.is_gzip, nthreads=self.nthreads)  # --gzip: BGZF, deflated on nthreads threads
            pos = 0
            with wf:
                while True:
                    item = self.q.get()
                    if item is None:
                        break
                    block, probs_t, labels_t, event = item[:4]
                    done = item[4] if len(item) > 4 else None   # hands the result slot back once its contents are formatted
                    if probs_t is None:   # a block without rows
                        if self.mark_blocks:
                            if self.is_gzip:
                                wf.end_block()
                            else:
                                self.block_ends.append(pos)
                        self.reader.release(block)
                        continue
                    event.synchronize()
                    if self.delay:   # DSP_WRITER_DELAY_MS (tests): a writer that cannot keep up with the GPU
                        import time as _time
                        _time.sleep(self.delay)
                    probs = probs_t.numpy()[:block.rows.n]
                    labels = labels_t.numpy()[:block.rows.n]
                    if self.canary:
                        self._check_live(block, probs, labels)
                    if self.n_vocab < 16 and block.rows.n and int(np.asarray(block.rows.kmer)[:block.rows.n].max()) >= self.n_vocab:
                        # a base code the embedding table does not hold: the reference's nn.Embedding raises this
                        # (models.py:186); the kernel clamps the index for memory safety, the run ends here
                        raise IndexError("index out of range in self")
                    text = textio.format_calls(block.rows, probs, labels, nthreads=self.nthreads)
                    wf.write(text)
                    if self.mark_blocks:
                        if self.is_gzip:
                            wf.end_block()
                        else:
                            pos += len(text)
                            self.block_ends.append(pos)
                    if self.freq is not None:
                        self.freq.add_block(block.rows, probs, labels)
                    if self.freq_dev is not None:
                        agg, side_stream = self.freq_dev
                        agg.add_block(block.rows, probs, labels, block.first_row, stream=side_stream)
                    self.rows += block.rows.n
                    if done is not None:
                        done()
                    self.reader.release(block)
            if self.mark_blocks and self.is_gzip:
                self.block_ends = list(wf.block_ends)   # complete once the writer is closed
        except BaseException as e:
            self.error = e
            while self.q.get() is not None:  # drain so the producer never blocks on a dead writer
                pass


def _call_mods_file(args, rank, local_rank, world):
    """One rank = one GPU: reader -> H2D -> forward -> D2H -> formatter -> part file."""
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    # control-plane collectives: tensors on this GPU over RCCL, on the host when the ranks had to fall back to gloo --
    # dist.comm_device decides, the collectives themselves are the same
    coll_dev = dev
    model = load_model(args, local_rank)
    mode, states_path = init_state_mode(args)
    file_states = FileInitStates(states_path, model) if mode == "file" else None
    _tick("model loaded")
    input_path = os.path.abspath(args.input_path)
    nthreads = dsp_dist.threads_per_rank(args.nproc)   # --nproc, capped by this rank's share of the node's CPUs

    # my byte range and the global index of my first row (plain text: a byte range of the file; BGZF .gz: a member range,
    # feed.FeatureReader._run_bgzf; a foreign single-stream .gz is inflated by every rank, which then knows all indices)
    first_row, byte_range, gz_ring, interleaved = 0, None, None, False
    counted = None   # the rows the counting pass (or the BGZF headers) promised this rank: checked against the rows parsed
    multi = dsp_dist.collective(world)   # several ranks (or DSP_FORCE_DIST=1: the same collectives with one)
    if input_path.endswith(".gz") and multi:
        mine = feed.count_rows_bgzf(input_path, world, rank, nthreads)
        if mine is not None:
            counted = mine
            counts = dsp_dist.all_gather_ints(mine, world, coll_dev)
            first_row = dsp_dist.exclusive_prefix(counts, rank)
        elif world > 1:
            # a foreign single-stream .gz (what the reference's `extract --gzip` writes): inflated ONCE per node by its
            # first rank into a shared-memory ring; every rank copies its own blocks out (feed.open_gz_ring).  Blocks are
            # dealt round-robin (block i -> rank i % world), so a rank's part file is not a contiguous piece of the
            # output: the writer remembers where each block's calls end and rank 0 interleaves the pieces again
            interleaved = True
            def gather(obj):   # (names / flags of the shared-memory ring: json through dist.comm_device, no pickled-object collective)
                return dsp_dist.all_gather_json(obj, world, coll_dev)
            gz_ring = feed.open_gz_ring(input_path, rank, world, int(os.environ.get("LOCAL_RANK", rank)),
                                        int(os.environ.get("LOCAL_WORLD_SIZE", world)), gather)
    if not input_path.endswith(".gz") and multi and not featfile.is_feature_file(input_path):
        import mmap
        size = os.path.getsize(input_path)
        if size:
            with open(input_path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
                byte_range = dsp_dist.byte_range_for_rank(mm, size, world, rank)
            mine = feed.count_rows_in_range(input_path, *byte_range, nthreads=nthreads)
            counted = mine
            counts = dsp_dist.all_gather_ints(mine, world, coll_dev)
            first_row = dsp_dist.exclusive_prefix(counts, rank)

    out_path = args.result_file
    if args.gzip and not out_path.endswith(".gz"):
        out_path += ".gz"  # call_modifications.py:264-267
    part_path = out_path if world == 1 else "%s.part%05d" % (out_path, rank)
    _remove_stale_parts(part_path, world)

    # --parse_on device (default): the reader only stages the text (one copy + row-start pass), the rows are parsed on the GPU
    # (csrc/dsp_parse_dev.hip) one block ahead of the forward; --parse_on host: this rank's parser threads, as before round 4
    device_parse = parse_on(args, nthreads) == "device"
    reader = feed.FeatureReader(input_path, args.seq_len, args.signal_len, rank=rank, world=world, nthreads=nthreads,
                                nbuf=5 if device_parse else 4, first_row=first_row, byte_range=byte_range, gz_ring=gz_ring,
                                device_parse=device_parse)
    dparse = None
    if reader.device_parse:
        from .parse_dev import DeviceRowParser
        dparse = DeviceRowParser(dev, args.seq_len, args.signal_len)
    freq, freq_dev = _make_freq(args, dev, world, nthreads)
    writer = _Writer(part_path, args.gzip, nthreads, reader, freq)
    writer.mark_blocks = interleaved
    writer.n_vocab = int(args.n_vocab)
    writer.class_num = int(args.class_num)
    cap = reader.cap
    _tick("reader built")
    model.reserve(cap)
    _tick("workspace reserved")
    reader.start()
    writer.start()
    stream = torch.cuda.current_stream(dev)
    copy_stream = torch.cuda.Stream(dev) if not os.environ.get("DSP_ONE_STREAM") else stream   # (DSP_ONE_STREAM=1: debugging aid)
    nout = 4
    out_probs = [torch.empty((cap, args.class_num), dtype=torch.float32, pin_memory=True) for _ in range(nout)]
    out_labels = [torch.empty((cap,), dtype=torch.uint8, pin_memory=True) for _ in range(nout)]
    # A result slot is refilled only after the WRITER has handed it back (it has formatted the slot's previous block) -- that
    # the copy into it has finished says nothing about the reader of the buffer.  (Until late in round 4 the slots went round
    # by block number, guarded by the copy's event: safe only while fewer blocks were in flight than slots, which the fifth
    # reader slot of the device parser broke at the end of a run with a writer slower than the GPU.)
    free_out = queue.Queue()
    slot_canary = canary.on()   # DSP_SLOT_CANARY=1: result slots are poisoned when handed back and verified when taken (canary.py)
    # DSP_TEST_RESULT_RING_BY_BLOCK=1 (tests only) re-enacts the round-4 bug: slots by block number, no hand-back awaited --
    # what the canary must catch without any timing help
    ring_by_block = int(os.environ.get("DSP_TEST_RESULT_RING_BY_BLOCK", "0") or 0)   # (its value: the length of that ring; round 4's was 4)
    for i in range(nout):
        if slot_canary:
            canary.poison([out_probs[i].numpy(), out_labels[i].numpy()])
        free_out.put(i)

    def hand_back(s_):
        if slot_canary:
            canary.poison([out_probs[s_].numpy(), out_labels[s_].numpy()])
        free_out.put(s_)

    def take_slot():
        """a result slot the writer has handed back; None when the writer is dead (the main loop ends the run with its error)"""
        while True:
            try:
                return free_out.get(timeout=0.2)
            except queue.Empty:
                if writer.error is not None:
                    return None
    k = 0
    n_rows = 0
    _tick("pinned output buffers")
    def issue(block, staged):
        """forward of one block (its inputs: uploaded from the host parser's pinned arrays, or parsed on the GPU), results
        towards the writer"""
        nonlocal k, n_rows, out_probs, out_labels
        rows = block.rows
        n = rows.n
        if n > out_probs[0].shape[0]:  # a block grew past the pinned output capacity: new slots, once the writer holds none of the old
            held = [take_slot() for _ in range(nout)]
            if None in held:
                return
            out_probs = [torch.empty((n, args.class_num), dtype=torch.float32, pin_memory=True) for _ in range(nout)]
            out_labels = [torch.empty((n,), dtype=torch.uint8, pin_memory=True) for _ in range(nout)]
            for i in held:
                if slot_canary:
                    canary.poison([out_probs[i].numpy(), out_labels[i].numpy()])
                free_out.put(i)
        if staged is not None:
            b, ev = staged
            ev.synchronize()   # (submitted a block ago: normally long done) the writer's small arrays and the flag count are here
            if int(block.slot["_torch"]["n_flagged"][0]) != 0:
                if os.environ.get("DSP_PARSE_DEBUG"):
                    st = b["status"][:n].cpu().numpy()
                    ro = b["row_off"][:n + 1].cpu().numpy()
                    sys.stderr.write("[parse_dev] flagged block first_row %d n %d n_bytes %d: status counts %s; row_off dev==host %s; first bad rows %s\n" % (
                        block.first_row, n, block.n_bytes, np.bincount(st, minlength=4).tolist(),
                        bool((ro == block.slot["_torch"]["row_off"][:n + 1].numpy()).all()), np.flatnonzero(st)[:8].tolist()))
                # rows outside the plain grammar: the host parser decides, and raises what the reference would
                dparse.host_fallback(rows, block.n_bytes, block.slot, b, nthreads, stream)
            else:
                stream.wait_event(ev)
            kmer, means, stds, lens, signals = b["kmer"][:n], b["means"][:n], b["stds"][:n], b["lens"][:n], b["signals"][:n]
        else:
            tt = block.slot.get("_torch")

            def dev_t(name):
                src = tt[name][:n] if tt is not None else torch.from_numpy(getattr(rows, name))
                return src.to(dev, non_blocking=True)
            # uploads go down their own stream so that block k+1's H2D runs under block k's forward
            with torch.cuda.stream(copy_stream):
                kmer, means, stds = dev_t("kmer"), dev_t("means"), dev_t("stds")
                lens, signals = dev_t("lens"), dev_t("signals")
                up_done = torch.cuda.Event()
                up_done.record(copy_stream)
            stream.wait_event(up_done)
            for t_in in (kmer, means, stds, lens, signals):
                t_in.record_stream(stream)
        model.site_offset = block.first_row
        _logits, probs, labels = model.forward(kmer, means, stds, lens, signals, want_labels=True,
                                               init_states=file_states.for_rows(block.first_row, n) if file_states else None)
        if freq_dev is not None:  # the calls go into the device-side call_freq records straight from HBM
            freq_dev.add_block(rows, probs, labels, block.first_row, stream=stream)
        slot = k % min(nout, ring_by_block) if ring_by_block else take_slot()
        if slot is None:
            return
        if slot_canary:
            canary.expect_poisoned([out_probs[slot].numpy(), out_labels[slot].numpy()], "main loop takes result slot %d for block %d" % (slot, k))
        ev_out = torch.cuda.Event()
        out_probs[slot][:n].copy_(probs, non_blocking=True)
        out_labels[slot][:n].copy_(labels, non_blocking=True)
        ev_out.record(stream)
        writer.q.put((block, out_probs[slot], out_labels[slot], ev_out, lambda s_=slot: hand_back(s_)))
        k += 1
        n_rows += n

    ahead = []   # device-parsed blocks whose parse has been submitted and whose forward has not: one block of lookahead
    for block in reader:
        if writer.error is not None:
            break
        if k == 0 and not ahead:
            _tick("first block parsed")
        if block.rows.n == 0:
            while ahead:
                issue(*ahead.pop(0))
            if interleaved:   # the piece table of the part file needs an (empty) entry for every block
                writer.q.put((block, None, None, None))
            else:
                reader.release(block)
            continue
        if block.n_bytes is not None:      # text staged by the reader: parse it on the GPU, under the previous block's forward
            ahead.append((block, dparse.submit(block.rows, block.n_bytes, block.slot, copy_stream)))
            if len(ahead) > 1:
                issue(*ahead.pop(0))
        else:
            issue(block, None)
    while ahead and writer.error is None:
        issue(*ahead.pop(0))
    _tick("last forward issued")
    if counted is not None and writer.error is None and n_rows - counted not in ((0, 1) if rank == world - 1 else (0,)):
        # the global row indices of the later ranks were derived from this count (they key the initial states): a wrong
        # one must not pass silently (a .gz whose headers carry counts that are not its rows, a file changed under the run)
        raise RuntimeError("rank %d parsed %d rows where the counting pass found %d: %s changed during the run, or the row "
                           "counts in its BGZF headers are not its own (DSP_BGZF_COUNT_BY_INFLATE=1 ignores them)"
                           % (rank, n_rows, counted, input_path))
    writer.q.put(None)
    writer.join()
    torch.cuda.synchronize(dev)
    _tick("writer joined")
    if writer.error is not None:
        raise writer.error
    _finish_freq(args, freq, freq_dev, rank, world)
    _tick("frequencies finished")
    if gz_ring is not None:
        gz_ring["ring"].close()
    if interleaved:
        np.asarray(writer.block_ends, np.int64).tofile(part_path + ".blocks")
    return n_rows, part_path, out_path, interleaved


def _make_freq(args, dev, world, nthreads):
    """--freq_file: (host aggregator, device aggregator).  Device (default): the calls stay in HBM as records and are
    reduced there, sharded over the ranks (call_mods_freq.DeviceSiteFrequency).  Host (--freq_on host): the native host
    table; with several ranks it is fed from the merged per-read file by rank 0."""
    if not getattr(args, "freq_file", None):
        return None, None
    from .call_mods_freq import DeviceSiteFrequency, SiteFrequency
    if getattr(args, "freq_on", "device") == "device":
        return None, DeviceSiteFrequency(args.prob_cf, dev, nthreads)
    return (SiteFrequency(args.prob_cf) if world == 1 else None), None


def _finish_freq(args, freq, freq_dev, rank, world):
    if freq_dev is not None:
        if not freq_dev.can_finish(world):   # collective.  Too many calls for the records to stay in HBM: the host table
            args._freq_from_file = True      # computes the same bytes from the per-read file once it is merged
            return
        table = freq_dev.finish(rank, world)  # collective: every rank takes part
        if table is not None:
            table.write(args.freq_file, args.freq_sort, args.freq_bed, args.gzip)
    elif freq is not None:
        freq.write(args.freq_file, args.freq_sort, args.freq_bed, args.gzip)


class _ReadsBlock(object):
    """What the writer needs from one batch of reads: the sites' sampleinfo rows + host k-mer codes."""
    __slots__ = ("rows", "first_row")


class _NoRelease(object):
    def release(self, block):
        pass


def _call_mods_reads(args, rank, local_rank, world):
    """The fast5-directory branch (call_modifications.py:559-583, :285-470): read records -> features extracted on the
    GPU (extract_features.py of this build) -> forward -> per-read calls, the features never leaving HBM.
    Input files: *.fast5 (native reader over the HDF5 C library) and *.reads.npz (reads.save_reads).  Files are dealt to ranks
    in contiguous ranges; output order = file order."""
    import torch
    from . import reads as dsp_reads
    from .extract_features import FeatureExtractor, _read_position_file
    from .utils.process_utils import get_contig2len
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if init_state_mode(args)[0] == "file":
        raise ValueError("--init_state file: names the rows of a feature file; a directory of reads has no row order to replay")
    model = load_model(args, local_rank)
    _tick("model loaded")
    files = dsp_reads.list_read_files(os.path.abspath(args.input_path), str2bool(args.recursively))
    if rank == 0:
        print("%d read files in total.." % len(files))
    lo, hi = (len(files) * rank) // world, (len(files) * (rank + 1)) // world
    files = files[lo:hi]
    chrom2len = get_contig2len(args.reference_path) if args.reference_path else None
    positions = _read_position_file(args.positions) if args.positions else None
    nthreads = dsp_dist.threads_per_rank(args.nproc)
    fx = FeatureExtractor(motifs=args.motifs, mod_loc=args.mod_loc, seq_len=args.seq_len, signal_len=args.signal_len,
                          normalize_method=args.normalize_method, chrom2len=chrom2len, positions=positions,
                          region=args.region, methy_label=1, is_dna=str2bool(args.is_dna), device=dev,
                          seed=getattr(args, "seed", 0), round_stats=False, nthreads=nthreads)
    out_path = args.result_file
    if args.gzip and not out_path.endswith(".gz"):
        out_path += ".gz"
    part_path = out_path if world == 1 else "%s.part%05d" % (out_path, rank)
    _remove_stale_parts(part_path, world)
    freq, freq_dev = _make_freq(args, dev, world, nthreads)
    writer = _Writer(part_path, args.gzip, nthreads, _NoRelease(), freq)
    writer.n_vocab = int(args.n_vocab)
    if freq_dev is not None:  # fed from the writer thread (the k-mer codes reach the host with the results)
        writer.freq_dev = (freq_dev, torch.cuda.Stream(dev))
    writer.start()

    # loader: files -> batches of reads, decoded by a thread pool ahead of the GPU work
    batches = dsp_reads.ReadBatches(files, max(1, int(args.f5_batch_size)) * 8, args.corrected_group,
                                    args.basecall_subgroup, first_file_index=lo, workers=min(8, nthreads),
                                    only_chrom=fx.regioninfo[0], procs=int(os.environ.get("DSP_READER_PROCS", "0")))  # decoding scales on the loader threads; reader processes are opt-in
    rq = queue.Queue(maxsize=3)

    def load():
        try:
            for item in batches:  # the host half of the extraction (concatenation into pinned staging, site strings)
                rq.put(fx.stage(item[0], read_uids=item[1]))   # runs here, under the previous batch's GPU work
        finally:
            rq.put(None)
    loader = threading.Thread(target=load, daemon=True)
    loader.start()

    stream = torch.cuda.current_stream(dev)
    n_rows, k = 0, 0
    ring = [dict(cap=0) for _ in range(6)]  # pinned result slots; a slot is refilled once the writer has handed it back
    free_ring = queue.Queue()
    for i in range(len(ring)):
        free_ring.put(i)
    slot_canary = canary.on()   # DSP_SLOT_CANARY=1 (canary.py): poisoned when handed back, verified when taken

    def slot_arrays(sl):
        return [sl[k].numpy() for k in ("probs", "labels", "kmer") if k in sl]

    def hand_back(i_):
        if slot_canary:
            canary.poison(slot_arrays(ring[i_]))
        free_ring.put(i_)
    fwd_chunk = 65536
    model.reserve(fwd_chunk)
    row_base = rank << 44  # row numbers for the output order only (rank-major = file order); the initial states of a
    # site are keyed by (read uid, base index in the read): ExtractedBatch.site_keys -- independent of ranks and batching
    while True:
        batch = rq.get()
        if batch is None:
            break
        if k == 0:
            _tick("first batch of reads staged")
        if writer.error is not None:
            fx.discard(batch)
            continue
        ext = fx.launch(batch, stream=stream)
        n = ext.n
        if n == 0:
            continue
        si = None
        while si is None:
            try:
                si = free_ring.get(timeout=0.2)
            except queue.Empty:
                if writer.error is not None:
                    break
        if si is None:        # the writer died: nothing comes back (its error ends the run below)
            continue
        slot = ring[si]
        k += 1
        if slot_canary and slot["cap"]:
            canary.expect_poisoned(slot_arrays(slot), "reads branch takes result slot %d for batch %d" % (si, k))
        if slot["cap"] < n:
            cap = n + n // 4
            slot.update(cap=cap, probs=torch.empty((cap, args.class_num), dtype=torch.float32, pin_memory=True),
                        labels=torch.empty((cap,), dtype=torch.uint8, pin_memory=True),
                        kmer=torch.empty((cap, args.seq_len), dtype=torch.uint8, pin_memory=True))
        h_probs, h_labels, h_kmer = slot["probs"], slot["labels"], slot["kmer"][:n]
        for a in range(0, n, fwd_chunk):  # forward in chunks of the reserved workspace size
            b = min(n, a + fwd_chunk)
            _logits, probs, labels = model.forward(ext.kmer[a:b], ext.means[a:b], ext.stds[a:b], ext.lens[a:b],
                                                   ext.signals[a:b], want_labels=True,
                                                   site_keys=ext.site_keys[a:b] if model.init_state == "randn" else None)
            h_probs[a:b].copy_(probs, non_blocking=True)
            h_labels[a:b].copy_(labels, non_blocking=True)
        h_kmer.copy_(ext.kmer, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream)
        ext.rows.kmer = h_kmer.numpy()  # valid once `ev` has passed; the writer waits on it first
        blk = _ReadsBlock()
        blk.rows = ext.rows
        blk.first_row = row_base + n_rows
        writer.q.put((blk, h_probs, h_labels, ev, lambda i_=si: hand_back(i_)))
        n_rows += n
    _tick("last forward issued")
    writer.q.put(None)
    writer.join()
    loader.join()
    _tick("writer joined")
    torch.cuda.synchronize(dev)
    if writer.error is not None:
        raise writer.error
    _finish_freq(args, freq, freq_dev, rank, world)
    print("%d of %d read files failed.." % (batches.failed, len(files)))  # :440
    return n_rows, part_path, out_path, False


def _remove_stale_parts(part_path, world):
    """What an earlier, aborted run with the same -o may have left behind: this rank's part file and its piece table.  The
    merge never looks at the disk to decide HOW to merge (the run tells it: `interleaved`), but a stale piece table next to
    a fresh part file must not survive to be trusted by anything else either."""
    if world == 1:
        return
    for p in (part_path, part_path + ".blocks"):
        try:
            os.remove(p)
        except OSError:
            pass


_BGZF_EOF = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def _merge_parts(out_path, world, interleaved=False, is_gzip=None):
    """concatenate the ranks' part files in rank order (gzip / BGZF members concatenate into a valid .gz; the empty
    end-of-file member of every part but the last is dropped so that the result is one well-formed BGZF file).
    interleaved: THIS run dealt the input's blocks round-robin (a foreign .gz) and every rank wrote its piece table --
    the run says so, the disk is not asked (a stale table of an aborted run must not choose the merge).
    is_gzip: the run's --gzip flag -- whether the parts ARE BGZF is what the run wrote, not what -o is called (a plain-text
    `-o calls.gz` without --gzip is plain text, as the reference writes it: ADVICE r4); None = the old guess from the name."""
    if is_gzip is None:
        is_gzip = out_path.endswith(".gz")
    parts = ["%s.part%05d" % (out_path, r) for r in range(world)]
    if interleaved:
        # interleaved sharding (a foreign .gz): piece k of rank r is block k * world + r of the input.  Pieces of a
        # --gzip part are whole BGZF members; the parts' end-of-file members are dropped and one is written at the end
        ends = [np.fromfile(p + ".blocks", np.int64) for p in parts]
        for p, e in zip(parts, ends):   # a piece table describes its part file up to the end-of-file member, or it is not its own
            want = os.path.getsize(p) - (len(_BGZF_EOF) if is_gzip else 0)
            if (int(e[-1]) if len(e) else 0) != want or (len(e) > 1 and bool(np.any(np.diff(e) < 0))):
                raise RuntimeError("%s.blocks does not describe %s (%d pieces ending at %d, file holds %d bytes of calls)"
                                   % (p, p, len(e), int(e[-1]) if len(e) else 0, want))
        files = [open(p, "rb") for p in parts]
        with open(out_path, "wb") as wf:
            for k in range(max(len(e) for e in ends)):
                for r in range(world):
                    if k < len(ends[r]):
                        a = int(ends[r][k - 1]) if k else 0
                        files[r].seek(a)
                        wf.write(files[r].read(int(ends[r][k]) - a))
            if is_gzip:
                wf.write(_BGZF_EOF)
        for f, p in zip(files, parts):
            f.close()
            os.remove(p)
            os.remove(p + ".blocks")
        return
    with open(out_path, "wb") as wf:
        for r in range(world):
            part = "%s.part%05d" % (out_path, r)
            size = os.path.getsize(part)
            with open(part, "rb") as rf:
                keep = size
                if is_gzip and r < world - 1 and size >= 28:
                    rf.seek(size - 28)
                    if rf.read(28) == _BGZF_EOF:
                        keep = size - 28
                    rf.seek(0)
                left = keep
                while left > 0:
                    chunk = rf.read(min(left, 16 << 20))
                    if not chunk:
                        break
                    wf.write(chunk)
                    left -= len(chunk)
            os.remove(part)


def _copy_range(src_fd, dst_fd, count, dst_off):
    """count bytes from the start of src_fd to dst_off of dst_fd: in the kernel where the file system allows it"""
    src_off, left = 0, count
    use_cfr = hasattr(os, "copy_file_range")
    while left > 0:
        n = 0
        if use_cfr:
            try:
                n = os.copy_file_range(src_fd, dst_fd, min(left, 1 << 30), src_off, dst_off)
            except OSError:   # EXDEV / EINVAL / ENOSYS: different file systems or a kernel without it
                use_cfr = False
                continue
            if n == 0:
                use_cfr = False
                continue
        else:
            chunk = os.pread(src_fd, min(left, 16 << 20), src_off)
            if not chunk:
                raise IOError("part file ended %d bytes early" % left)
            n = os.pwrite(dst_fd, chunk, dst_off)
        src_off += n
        dst_off += n
        left -= n


def _merge_parts_by_all_ranks(out_path, part_path, rank, world, coll_dev, interleaved=False, is_gzip=None):
    """Collective.  The per-read calls of N ranks become one file without funnelling them through rank 0: the ranks agree
    on the sizes (one all_gather), rank 0 sizes the result, and EVERY rank copies its own part to its offset at the same
    time (config 5's shape: 1 G rows = 60 GB of calls; one process copying them was a quarter of the run, with seven GPUs
    idle behind the barrier).  --gzip: the empty end-of-file member of every part but the last is left out, so that the
    result is one well-formed BGZF file.  The interleaved pieces of a foreign .gz input keep the rank-0 merge."""
    import torch.distributed as dist
    if is_gzip is None:
        is_gzip = out_path.endswith(".gz")
    if interleaved:   # what the run did, the same on every rank -- not what lies on the disk
        dist.barrier()
        if rank == 0:
            _merge_parts(out_path, world, True, is_gzip)
        return
    size = os.path.getsize(part_path)
    keep = size
    if is_gzip and rank < world - 1 and size >= 28:
        with open(part_path, "rb") as rf:
            rf.seek(size - 28)
            if rf.read(28) == _BGZF_EOF:
                keep = size - 28
    keeps = dsp_dist.all_gather_ints(keep, world, coll_dev)
    if rank == 0:
        with open(out_path, "wb") as wf:
            wf.truncate(sum(keeps))
    dist.barrier()
    src = os.open(part_path, os.O_RDONLY)
    dst = os.open(out_path, os.O_WRONLY)
    try:
        _copy_range(src, dst, keep, sum(keeps[:rank]))
    finally:
        os.close(src)
        os.close(dst)
    os.remove(part_path)


def _self_launch(args, argv=None):
    """The reference starts --nproc_gpu model processes itself and deals them to the visible GPUs round-robin
    (call_modifications.py:523-529, :613-621).  Here: when call_mods is started plainly (no torch.distributed
    launcher) on a node with several GPUs and --nproc_gpu > 1, it starts min(nproc_gpu, GPUs) ranks of itself under
    torch.distributed.run as a CHILD process and returns its exit code; None = carry on in this process.  Like
    bench.py's launcher the parent makes no GPU call at all -- it does not even import torch: the GPUs are counted from the
    ROCm device filters or the KFD topology (dist.visible_gpu_count) -- because a process that has initialised the GPU must
    never be the one that starts replacing itself."""
    import socket
    import subprocess
    if "RANK" in os.environ or "WORLD_SIZE" in os.environ or os.environ.get("DSP_NO_SELF_LAUNCH"):
        return None
    want = int(getattr(args, "nproc_gpu", 1) or 1)
    if want <= 1:
        return None
    ngpu = dsp_dist.visible_gpu_count()   # environment filters / KFD topology / a fresh child: this process never loads HIP here
    n = min(want, ngpu)
    if n <= 1:
        return None
    argv = list(sys.argv[1:] if argv is None else argv)
    if argv and argv[0] == "call_mods":
        argv = argv[1:]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), "-m", "deepsignal_plant_amd.deepsignal_plant", "call_mods"] + argv
    print("[main] %d GPUs visible, --nproc_gpu %s: starting %d ranks (one per GPU)" % (ngpu, args.nproc_gpu, n))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def call_mods(args):
    """Main function of calling modifications (mirror of call_modifications.py:532-640)."""
    rc = _self_launch(args)
    if rc is not None:
        if rc != 0:
            raise RuntimeError("call_mods: the multi-GPU child run failed with exit code %d" % rc)
        return None
    print("[main] call_mods starts..")
    start = time.time()
    del _TICKS[:]
    _tick("start")
    feed.refresh_env()   # (DSP_BLOCK_BYTES: a second call_mods in one process may run under other settings)
    import torch
    from . import _native
    _native.lib()  # fail loudly before any work if the HIP library is missing
    print("cuda availability: {}".format(torch.cuda.is_available()))

    init_state_mode(args)   # a bad --init_state ends the run before anything is loaded
    model_path = os.path.abspath(args.model_path)
    if not os.path.exists(model_path):
        raise ValueError("--model_path is not set right!")  # :550-551
    input_path = os.path.abspath(args.input_path)
    if not os.path.exists(input_path):
        raise ValueError("--input_path does not exist!")  # :553-554
    if not torch.cuda.is_available():
        raise RuntimeError("no MI355X visible: this build has no CPU path")

    rank, local_rank, world = dsp_dist.env_world()
    ndev = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    # placement first: the page-locked slots and the staging / writer threads created below then sit next to this rank's
    # GPU (numa by default with several ranks; DSP_RANK_AFFINITY=off opts out -- dist.pin_rank)
    dsp_dist.place_rank(rank, local_rank, local_world, ndev)
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    # one process per GPU over RCCL (gloo when ranks have to share GPUs; DSP_FORCE_DIST=1: a one-rank RCCL group)
    dsp_dist.init_process_group(world, rank, local_rank, ndev)
    dist_on = dsp_dist.collective(world)
    _tick("imports, checks, process group")
    if os.path.isdir(input_path):  # reads in, calls out: extraction + forward on the GPU (:559-583)
        n_rows, part_path, out_path, interleaved = _call_mods_reads(args, rank, local_rank, world)
    else:
        n_rows, part_path, out_path, interleaved = _call_mods_file(args, rank, local_rank, world)
    if dist_on:
        import torch.distributed as dist
        cdev = torch.device("cuda", local_rank)   # (dist.comm_device maps it to the host under gloo)
        total = sum(dsp_dist.all_gather_ints(n_rows, world, cdev))
        dist.barrier()
        _tick("all ranks done")
    if world > 1:
        _merge_parts_by_all_ranks(out_path, part_path, rank, world, cdev, interleaved, bool(args.gzip))
        dist.barrier()
        if rank == 0:
            if getattr(args, "freq_file", None) and (getattr(args, "freq_on", "device") == "host" or
                                                     getattr(args, "_freq_from_file", False)):
                # --freq_on host with several ranks: aggregate the merged per-read calls in file order (the default,
                # --freq_on device, has already reduced them on the GPUs: DeviceSiteFrequency.finish)
                from .call_mods_freq import SiteFrequency
                agg = SiteFrequency(args.prob_cf)
                agg.add_calls_file(out_path)
                agg.write(args.freq_file, args.freq_sort, args.freq_bed, args.gzip)
        dist.barrier()
    else:
        total = n_rows if not dist_on else total
        if getattr(args, "_freq_from_file", False):
            from .call_mods_freq import SiteFrequency
            agg = SiteFrequency(args.prob_cf)
            agg.add_calls_file(out_path)
            agg.write(args.freq_file, args.freq_sort, args.freq_bed, args.gzip)
    if dist_on and not os.environ.get("DSP_KEEP_PROCESS_GROUP"):   # (the tests' job runner runs several call_mods in one launch)
        import torch.distributed as dist
        dist.destroy_process_group()
    if rank == 0:
        dt = time.time() - start
        if _TICKS:
            _tick("merged / finished")
            print("[call_mods] seconds at: " + ", ".join("%s %.2f" % (k, t - start) for k, t in _TICKS[1:]), file=sys.stderr)
        print("[main] call_mods costs %.2f seconds.." % dt)
        print("[main] %d sites on %d GPU(s): %.0f sites/s" % (total, world, total / max(dt, 1e-9)))
    return total


def add_call_mods_args(p):
    """The reference's call_mods flag surface (deepsignal_plant.py:204-316 == call_modifications.py:643-765)
    plus the build-only flags --init_state / --seed."""
    g = p.add_argument_group("INPUT")
    g.add_argument("--input_path", "-i", type=str, required=True,
                   help="feature file written by `extract` (plain or .gz), its binary form written by `pack_features` "
                        "(.dspf), or a directory of reads (*.fast5 / *.reads.npz) to extract features from on the fly")
    g.add_argument("--f5_batch_size", type=int, default=30,
                   help="reads per reader batch in the reference (default 30); this build loads 8x as many per GPU batch, results do not depend on it")
    g = p.add_argument_group("CALL")
    g.add_argument("--model_path", "-m", type=str, required=True, help="trained model checkpoint (.ckpt, a state_dict)")
    g.add_argument("--model_type", type=str, default="both_bilstm", choices=["both_bilstm", "seq_bilstm", "signal_bilstm"])
    g.add_argument("--seq_len", type=int, default=13, help="k-mer length, default 13")
    g.add_argument("--signal_len", type=int, default=16, help="signals per base, default 16")
    g.add_argument("--layernum1", type=int, default=3, help="BiLSTM layers on the combined feature, default 3")
    g.add_argument("--layernum2", type=int, default=1, help="BiLSTM layers on seq / signal features, default 1")
    g.add_argument("--class_num", type=int, default=2)
    g.add_argument("--dropout_rate", type=float, default=0)
    g.add_argument("--n_vocab", type=int, default=16, help="base vocabulary size (IUPAC)")
    g.add_argument("--n_embed", type=int, default=4, help="base embedding size")
    g.add_argument("--is_base", type=str, default="yes", help="use base features in the seq model, default yes")
    g.add_argument("--is_signallen", type=str, default="yes", help="use per-base signal length in the seq model, default yes")
    g.add_argument("--batch_size", "-b", type=int, default=512,
                   help="batch size of the reference (default 512); this build batches whole parsed blocks, results do not depend on it")
    g.add_argument("--hid_rnn", type=int, default=256, help="BiLSTM hidden size for the combined feature")
    g = p.add_argument_group("OUTPUT")
    g.add_argument("--result_file", "-o", type=str, required=True, help="per-read call file to write")
    g.add_argument("--gzip", action="store_true", default=False, help="gzip the output")
    g = p.add_argument_group("EXTRACTION (used when --input_path is a directory of *.fast5 / *.reads.npz files)")
    g.add_argument("--recursively", "-r", type=str, default="yes")
    g.add_argument("--corrected_group", type=str, default="RawGenomeCorrected_000")
    g.add_argument("--basecall_subgroup", type=str, default="BaseCalled_template")
    g.add_argument("--is_dna", type=str, default="yes")
    g.add_argument("--normalize_method", type=str, choices=["mad", "zscore"], default="mad")
    g.add_argument("--methy_label", type=int, choices=[1, 0], default=1, help=argparse.SUPPRESS)  # commented out in the
    # reference's call_mods parsers (deepsignal_plant.py:280-283, call_modifications.py:721); accepted and ignored here so
    # that a command line carrying it (it is an `extract` flag, deepsignal_plant.py:150) still runs
    g.add_argument("--motifs", type=str, default="CG")
    g.add_argument("--mod_loc", type=int, default=0)
    g.add_argument("--region", type=str, default=None)
    g.add_argument("--positions", type=str, default=None)
    g.add_argument("--reference_path", type=str, default=None)
    p.add_argument("--nproc", "-p", type=int, default=10, help="host threads for parsing/formatting, default 10")
    p.add_argument("--nproc_gpu", type=int, default=2,
                   help="model processes in the reference (default 2); here: GPUs to use -- started plainly on a node "
                        "with several GPUs, call_mods runs min(nproc_gpu, GPUs) ranks of itself, one per GPU")
    g = p.add_argument_group("MI355X build")
    g.add_argument("--init_state", type=str, default="randn",
                   help="LSTM initial states: 'randn' = N(0,1) like the reference's init_hidden (in-kernel Philox), 'zeros', or "
                        "'file:<states.npz>' = explicit states of every input row in init_hidden's layout (h_seq, c_seq, h_sig, "
                        "c_sig, h_comb, c_comb; replays a captured reference run)")
    g.add_argument("--seed", type=int, default=0, help="seed of the in-kernel initial-state generator")
    g.add_argument("--parse_on", type=str, default=None, choices=["device", "host", "auto"],
                   help="where the feature rows are parsed: 'device' (default) = the host only stages the text, one GPU thread per "
                        "token parses it (rows outside the plain grammar, and every error, still go through the host parser); "
                        "'host' = this rank's --nproc parser threads.  Same values either way")
    g.add_argument("--precision", type=str, default=None, choices=["fp32", "bf16x6", "bf16x9", "fp16x3"],
                   help="how the fp32 products of the combined BiLSTM stack are evaluated: fp32 matrix cores (default), or "
                        "split into low-precision pieces on the fast matrix pipes with fp32 accumulation (bf16x9: nine bf16 "
                        "piece products, exact; bf16x6: without the three smallest, 1.5x; fp16x3: two fp16 pieces, three "
                        "products, 2.4x; probabilities within 1e-6 of the fp32 path)")
    g.add_argument("--freq_file", type=str, default=None,
                   help="also write the per-site modification frequency (what `call_freq` computes from the result file) "
                        "without re-reading the per-read calls")
    g.add_argument("--freq_on", type=str, default="device", choices=["device", "host"],
                   help="where --freq_file is aggregated: 'device' (default) keeps the calls in HBM as records, deals them to "
                        "the ranks by site (one RCCL all_to_all) and reduces them there; 'host' = the native host table "
                        "(with several ranks: rank 0 re-reads the merged per-read file).  Same bytes either way")
    g.add_argument("--prob_cf", type=float, default=0.5, help="call_freq threshold on abs(prob1-prob0), default 0.5")
    g.add_argument("--freq_bed", action="store_true", default=False, help="--freq_file in bedMethyl format")
    g.add_argument("--freq_sort", action="store_true", default=False, help="sort --freq_file by chromosome and position")
    return p


def main():
    parser = add_call_mods_args(argparse.ArgumentParser("call modifications"))
    args = parser.parse_args()
    display_args(args)
    call_mods(args)


if __name__ == '__main__':
    sys.exit(main())
