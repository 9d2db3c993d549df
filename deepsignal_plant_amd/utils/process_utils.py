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
