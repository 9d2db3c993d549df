"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's per-read feature extraction, the stage
that produces the rows call_mods consumes.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this; the product path (deepsignal_plant_amd/) never does.

Follows deepsignal_plant/extract_features.py:
  _rescale_signals      :273-274   pA = scaling * (raw + offset), float64
  _normalize_signals    :179-190   mad: (x - median) / mad, zscore: (x - mean) / std; np.around(.., 6)
  _get_signals_rect     :232-251   per base: round 6, centred zero padding (left = pad // 2) or a sorted random
                                   sample of signals_len values when the base is longer
  _extract_features     :277-378   event slicing, motif sites, +/- strand coordinates, region / positions filters,
                                   k-mer window, per-base mean / std / length
  _features_to_str      :381-395   the feature-TSV row
and utils/process_utils.py:97-112 (get_refloc_of_methysite_in_motif).  `robust.mad` is statsmodels'
(robust/scale.py: median(|a - median(a)| / c), c = norm.ppf(3/4)), restated here.  Pinned twice: by
tests/golden/f6_extract.npz (the imported reference with its three HDF5 accessors replaced and the same restated mad) and
by tests/golden/fast5/ (F7: the reference UNMODIFIED, h5py + statsmodels' own robust.mad from the image's python3.9, on
real fast5 files -- tests/test_fast5_reader.py compares this module's rows with those byte for byte).

Sampling of long bases: the reference calls the process-global, unseeded `random.sample`, i.e. it is not
reproducible against itself.  sampler="python" uses the same `random.sample` calls in the same order (pins the
restatement against a reference run made under random.seed); sampler="hash" is the product's deterministic
counter-based sampler (Floyd's subset sampling over a 64-bit mix of (seed, read uid, base index, step))."""
from __future__ import annotations

import random

import numpy as np

MAD_C = 0.6744897501960817  # scipy.stats.norm.ppf(0.75), statsmodels' default c
KEY_SEP = "||"  # extract_features.py:40
M64 = (1 << 64) - 1


def mad(a):
    """statsmodels.robust.mad(a) with its defaults (restated; pinned by F7, see the module docstring)."""
    a = np.asarray(a)
    center = np.median(a)
    err = np.abs(a - center) / MAD_C
    return np.median(err)


def rescale_signals(raw, scaling, offset):
    return np.array(scaling * (raw + offset), dtype=float)  # :273-274


def normalize_signals(signals, normalize_method="mad"):
    if normalize_method == "zscore":
        sshift, sscale = np.mean(signals), float(np.std(signals))
    elif normalize_method == "mad":
        sshift, sscale = np.median(signals), float(mad(signals))
    else:
        raise ValueError("")
    if sscale == 0.0:
        norm = signals
    else:
        norm = (signals - sshift) / sscale
    return np.around(norm, decimals=6)


def mix64(x):
    """splitmix64 finaliser on Python ints (mod 2^64)."""
    x &= M64
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & M64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & M64
    x ^= x >> 31
    return x


def hash_sample_sorted(n, k, seed, read_uid, base_index):
    """k of range(n), ascending, uniform over all subsets: Floyd's algorithm driven by a counter-based 32-bit
    stream -- step q draws v uniformly from [0, n-k+q] as floor(r_q * (n-k+q+1) / 2^32) and takes n-k+q instead
    when v is already in the sample."""
    h = mix64((seed ^ ((read_uid * 0x9E3779B97F4A7C15) & M64)) + ((base_index * 0xD1B54A32D192ED03) & M64))
    out = []
    for q in range(k):
        jj = n - k + q
        r = mix64(h + q) >> 32
        v = (r * (jj + 1)) >> 32
        out.append(jj if v in out else v)
    return sorted(out)


def get_signals_rect(signals_list, signals_len=16, sampler="python", seed=0, read_uid=0, first_base=0):
    rect = []
    for j, s in enumerate(signals_list):
        s = list(np.around(s, decimals=6))
        if len(s) < signals_len:
            pad = signals_len - len(s)
            left = pad // 2
            s = [0.] * left + s + [0.] * (pad - left)
        elif len(s) > signals_len:
            if sampler == "python":
                idx = sorted(random.sample(range(len(s)), signals_len))
            else:
                idx = hash_sample_sorted(len(s), signals_len, seed, read_uid, first_base + j)
            s = [s[x] for x in idx]
        rect.append(s)
    return rect


def motif_sites(seq, motifset, methyloc=0):
    """utils/process_utils.py:97-112"""
    motifset = set(motifset)
    mlen = len(next(iter(motifset)))
    return [i + methyloc for i in range(0, len(seq) - mlen + 1) if seq[i:i + mlen] in motifset]


def extract_features(reads, normalize_method, motif_seqs, methyloc, chrom2len, kmer_len, signals_len, methy_label,
                     positions=None, regioninfo=(None, None, None), sampler="python", seed=0, first_read_uid=0,
                     read_uids=None):
    """reads: iterable of objects with the ReadRecord attributes (deepsignal_plant_amd/reads.py documents them;
    only attribute access is used).  Returns the reference's features_list (tuples, :370-372)."""
    if kmer_len % 2 == 0:
        raise ValueError("kmer_len must be odd")
    nb = (kmer_len - 1) // 2
    out = []
    rg_chrom, rg_start, rg_end = regioninfo
    for ridx, rd in enumerate(reads):
        chrom, chrom_start, alignstrand = rd.chrom, rd.chrom_start, rd.alignstrand
        if rg_chrom is not None and rg_chrom != chrom:
            continue
        raw = rescale_signals(rd.raw, rd.scaling, rd.offset)
        norm = normalize_signals(raw, normalize_method)
        seq = rd.ev_base.tobytes().decode()
        signal_list = [norm[int(s):int(s) + int(ln)] for s, ln in zip(rd.ev_start, rd.ev_len)]
        read_rg_start = chrom_start if rg_start is None else rg_start
        read_rg_end = chrom_start + len(seq) if rg_end is None else rg_end
        if read_rg_start >= chrom_start + len(seq) or read_rg_end <= chrom_start:
            continue
        chromlen = chrom2len.get(chrom) if chrom2len is not None else None
        for loc in motif_sites(seq, set(motif_seqs), methyloc):
            if not (nb <= loc < len(seq) - nb):
                continue
            if alignstrand == "-":
                pos = chrom_start + len(seq) - 1 - loc
                pos_in_strand = chromlen - 1 - pos if chromlen is not None else -1
            else:
                pos = chrom_start + loc
                pos_in_strand = pos if chromlen is not None else -1
            if rg_chrom is not None and (pos < read_rg_start or pos >= read_rg_end):
                continue
            if positions is not None and KEY_SEP.join([chrom, str(pos), alignstrand]) not in positions:
                continue
            k_mer = seq[loc - nb:loc + nb + 1]
            k_signals = signal_list[loc - nb:loc + nb + 1]
            lens = [len(x) for x in k_signals]
            means = [np.mean(x) for x in k_signals]
            stds = [np.std(x) for x in k_signals]
            uid = read_uids[ridx] if read_uids is not None else first_read_uid + ridx
            rect = get_signals_rect(k_signals, signals_len, sampler, seed, uid, loc - nb)
            out.append((chrom, pos, alignstrand, pos_in_strand, rd.readname, rd.strand, k_mer, means, stds, lens,
                        rect, methy_label))
    return out


def features_to_str(features):
    chrom, pos, alignstrand, pos_in_strand, readname, strand, k_mer, means, stds, lens, rect, label = features
    return "\t".join([chrom, str(pos), alignstrand, str(pos_in_strand), readname, strand, k_mer,
                      ",".join(str(x) for x in np.around(means, decimals=6)),
                      ",".join(str(x) for x in np.around(stds, decimals=6)),
                      ",".join(str(x) for x in lens),
                      ";".join(",".join(str(y) for y in x) for x in rect), str(label)])


def features_to_arrays(features_list, kmer_len, signals_len, round_stats):
    """The tensors the model is fed from a features_list: float32 narrowing of the float64 features, as
    FloatTensor does (call_modifications.py:159-162).  round_stats=True is the TSV route (means/stds rounded to 6
    decimals by _features_to_str before they reach the model), False the direct fast5 route
    (call_modifications.py:285-325, which hands the unrounded values over)."""
    code = {c: i for i, c in enumerate("ACGTNWSMKRYBVDHZ")}
    n = len(features_list)
    kmer = np.zeros((n, kmer_len), np.uint8)
    means = np.zeros((n, kmer_len), np.float32)
    stds = np.zeros((n, kmer_len), np.float32)
    lens = np.zeros((n, kmer_len), np.int32)
    signals = np.zeros((n, kmer_len, signals_len), np.float32)
    info = []
    for i, f in enumerate(features_list):
        info.append("\t".join([f[0], str(f[1]), f[2], str(f[3]), f[4], f[5]]))
        kmer[i] = [code[c] for c in f[6]]
        m, s = np.asarray(f[7], np.float64), np.asarray(f[8], np.float64)
        if round_stats:
            m, s = np.around(m, decimals=6), np.around(s, decimals=6)
        with np.errstate(over="ignore"):
            means[i], stds[i] = m.astype(np.float32), s.astype(np.float32)
        lens[i] = f[9]
        signals[i] = np.asarray(f[10], np.float64).astype(np.float32)
    return dict(sampleinfo=info, kmer=kmer, means=means, stds=stds, lens=lens, signals=signals,
                labels=np.array([f[11] for f in features_list], np.int32))
