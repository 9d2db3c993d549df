"""Alphabet / flag helpers of the call_mods path (mirror of deepsignal_plant/utils/process_utils.py:25-29,
:51-56, :394-401)."""

# 16-symbol IUPAC code table, process_utils.py:25-29
base2code_dna = {'A': 0, 'C': 1, 'G': 2, 'T': 3, 'N': 4, 'W': 5, 'S': 6, 'M': 7, 'K': 8, 'R': 9, 'Y': 10,
                 'B': 11, 'V': 12, 'D': 13, 'H': 14, 'Z': 15}
code2base_dna = dict((v, k) for k, v in base2code_dna.items())

CODE2BASE_STR = "".join(code2base_dna[i] for i in range(16))

nproc_to_call_mods_in_cpu_mode = 2  # process_utils.py:51 (kept for CLI compatibility; unused on the GPU path)


def str2bool(v):
    # process_utils.py:54-56
    return v.lower() in ("yes", "true", "t", "1")


def display_args(args):
    """Echo the parsed flags in the reference's layout (process_utils.py:394-401)."""
    bar = "# " + "=" * 47
    lines = [bar, "## parameters: "]
    lines += ["%s:\n\t%s" % (k, v) for k, v in vars(args).items() if k != "func"]
    lines.append(bar)
    print("\n".join(lines))


# IUPAC expansion of --motifs (process_utils.py:37-48, :115-145): "CHG" -> CAG, CCG, CTG
_IUPAC_DNA = {'A': 'A', 'T': 'T', 'C': 'C', 'G': 'G', 'R': 'AG', 'M': 'AC', 'S': 'CG', 'Y': 'CT', 'K': 'GT', 'W': 'AT',
              'B': 'CGT', 'D': 'AGT', 'H': 'ACT', 'V': 'ACG', 'N': 'ACGT'}


def get_motif_seqs(motifs, is_dna=True):
    """Comma-separated IUPAC motifs -> list of explicit sequences, in the reference's order (first position
    varies slowest).  A letter outside the IUPAC table raises KeyError like the reference."""
    import itertools
    table = _IUPAC_DNA if is_dna else {k.replace('T', 'U'): v.replace('T', 'U') for k, v in _IUPAC_DNA.items()}
    seqs = []
    for motif in motifs.strip().split(','):
        choices = [table[c] for c in motif.strip().upper()]
        seqs += ["".join(p) for p in itertools.product(*choices)]
    return seqs


def parse_region_str(regionstr):
    """"chrom", "chrom:start" or "chrom:start-end" (0-based, half open) -> (chrom, start, end); process_utils.py:164-187"""
    if regionstr is None:
        return None, None, None
    try:
        region = regionstr.strip()
        if ":" not in region:
            return region, None, None
        chrom, span = region.split(":")
        if "-" in span:
            start, end = span.split("-")
            return chrom, int(start), int(end)
        return chrom, int(span), None
    except Exception:
        raise ValueError("--region not set right!")


def get_contig2len(ref_path):
    """FASTA -> {contig name (up to the first blank): sequence length} (utils/ref_reader.py:7-13, :41-62)."""
    out, name, n = {}, None, 0
    with open(ref_path, "r") as rf:
        for line in rf:
            if line.startswith(">"):
                if name is not None:
                    out[name] = n
                name, n = line.strip()[1:].split(" ")[0], 0
            else:
                n += len(line.strip())
    if name is not None:
        out[name] = n
    return out
