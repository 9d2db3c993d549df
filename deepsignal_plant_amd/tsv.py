"""Synthetic feature-TSV rows in the exact format the reference's extractor writes
(deepsignal_plant/extract_features.py:381-395 `_features_to_str`, :232-251 `_get_signals_rect`):
12 tab-separated fields -- chrom, pos, strand, pos_in_strand, readname, read_strand, k_mer, means(csv),
stds(csv), lens(csv), signals(';'-separated groups of csv), label.  Floats are np.around(x, 6) printed with
str(); short bases are zero-padded centred (left = pad // 2)."""
from __future__ import annotations

import numpy as np

_BASES = "ACGT"


def synth_rows(n, seq_len=13, signal_len=16, seed=0, sites_per_read=50, n_chroms=5, wide_alphabet=False,
               first_index=0, n_sites=None):
    """Yield n text rows (no trailing newline), SURVEY.md 8(d) statistics.  n_sites: the rows cycle over that many
    genome sites (row i calls site i % n_sites, from read i // sites_per_read), i.e. every site is covered by
    n / n_sites different reads -- what call_freq aggregates; default: every row its own site."""
    rng = np.random.default_rng(seed)
    alphabet = "ACGTNWSMKRYBVDHZ" if wide_alphabet else _BASES
    for i in range(first_index, first_index + n):
        read = i // sites_per_read
        chrom = "chr%d" % (read % n_chroms + 1)
        pos = 1000 + 7 * i
        strand = "+" if read % 2 == 0 else "-"
        if n_sites:
            j = i % n_sites
            chrom = "chr%d" % ((j // sites_per_read) % n_chroms + 1)
            pos = 1000 + 7 * j
            strand = "+" if (j // sites_per_read) % 2 == 0 else "-"
        kmer = [alphabet[j] for j in rng.integers(0, len(alphabet), size=seq_len)]
        kmer[seq_len // 2] = "C"
        means = np.around(rng.standard_normal(seq_len), decimals=6)
        stds = np.around(np.abs(rng.normal(0.25, 0.1, size=seq_len)), decimals=6)
        lens = rng.integers(2, 40, size=seq_len)
        groups = []
        for b in range(seq_len):
            ln = int(min(lens[b], signal_len))
            sig = [float(x) for x in np.around(rng.standard_normal(ln), decimals=6)]
            pad = signal_len - ln
            left = pad // 2
            sig = [0.] * left + sig + [0.] * (pad - left)
            groups.append(",".join(str(y) for y in sig))
        yield "\t".join([chrom, str(pos), strand, str(pos + 3 if strand == "+" else 30000000 - pos),
                         "read_%06d" % read, "t", "".join(kmer),
                         ",".join(str(x) for x in means), ",".join(str(x) for x in stds),
                         ",".join(str(x) for x in lens), ";".join(groups), str(i % 2)])


def write_tsv(path, n, **kw):
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "wt") as f:
        for row in synth_rows(n, **kw):
            f.write(row + "\n")
    return path
