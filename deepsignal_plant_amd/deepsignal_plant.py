#!/usr/bin/env python
"""`deepsignal_plant` command line (mirror of deepsignal_plant/deepsignal_plant.py:85-481).

Implemented natively: ``call_mods`` (feature file or reads in, per-read calls out -- the hot path), ``call_freq``
(per-read calls in, per-site frequency out -- SURVEY.md 8(f) next-1), ``extract`` (reads in, feature rows out --
next-3) and the build-only ``pack_features`` (feature TSV -> binary container, next-2).  ``train`` and ``denoise``
are registered so that scripts see the same command set, and exit with a clear message (out of scope
for this build)."""
from __future__ import absolute_import

import argparse
import sys

from ._version import VERSION
from .utils.process_utils import display_args


def main_call_mods(args):
    from .call_modifications import call_mods  # lazy, like deepsignal_plant.py:50-54
    display_args(args)
    call_mods(args)


def main_call_freq(args):
    from .call_mods_freq import call_mods_frequency_to_file
    display_args(args)
    call_mods_frequency_to_file(args)


def main_extract(args):
    from .extract_features import extract_features
    display_args(args)
    extract_features(args)


def main_pack_features(args):
    from .featfile import pack_features
    display_args(args)
    n = pack_features(args.input_path, args.result_file, args.seq_len, args.signal_len, args.block_rows, args.nproc)
    print("[main] pack_features: %d rows -> %s" % (n, args.result_file))


def _not_in_this_build(name):
    def run(_args):
        sys.stderr.write("deepsignal_plant %s: not part of the MI355X call_mods build (use the reference "
                         "implementation for this step)\n" % name)
        sys.exit(2)
    return run


def main():
    parser = argparse.ArgumentParser(prog="deepsignal_plant",
                                     description="detecting base modifications from Nanopore sequencing reads of plants, "
                                                 "MI355X-native call_mods path",
                                     formatter_class=argparse.RawTextHelpFormatter)
    parser.add_argument("-v", "--version", action="version", version="deepsignal-plant_amd version: {}".format(VERSION))
    sub = parser.add_subparsers(title="modules", help="deepsignal_plant modules, use -h/--help for help")
    from .call_modifications import add_call_mods_args
    sub_call_mods = sub.add_parser("call_mods", description="call modifications")
    add_call_mods_args(sub_call_mods)
    sub_call_mods.set_defaults(func=main_call_mods)
    from .call_mods_freq import add_call_freq_args
    sub_call_freq = sub.add_parser("call_freq", description="call frequency from the per-read call file(s) of call_mods")
    add_call_freq_args(sub_call_freq)
    sub_call_freq.set_defaults(func=main_call_freq)
    from .featfile import add_pack_features_args
    sub_pack = sub.add_parser("pack_features", description="feature TSV -> binary feature container (.dspf) that "
                                                           "call_mods reads without parsing (build-only helper)")
    add_pack_features_args(sub_pack)
    sub_pack.set_defaults(func=main_pack_features)
    from .extract_features import add_extract_args
    sub_extract = sub.add_parser("extract", description="extract features from resquiggled reads (on the GPU)")
    add_extract_args(sub_extract)
    sub_extract.set_defaults(func=main_extract)
    for name in ("train", "denoise"):
        sp = sub.add_parser(name, description="%s (not part of this build)" % name, add_help=True)
        sp.add_argument("rest", nargs=argparse.REMAINDER)
        sp.set_defaults(func=_not_in_this_build(name))
    args = parser.parse_args()
    if hasattr(args, "func"):
        args.func(args)
    else:
        parser.print_help()


if __name__ == '__main__':
    sys.exit(main())
