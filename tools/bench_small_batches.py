#!/usr/bin/env python3
"""Small batches through dsp_forward: sites/s of K handles issuing concurrently on K streams.

One forward of up to 8,192 sites occupies at most one workgroup per CU and takes the same 6.6 ms whatever its size
(tools/batch_sweep.sh), so a caller that keeps the reference's batch of 512 (call_modifications.py:147) leaves 15 of 16
CUs idle.  include/dsp_amd.h's answer is one handle per stream: every handle owns its own scratch, the repacked weights
are 19 MB, and forwards of different handles overlap on the GPU.  This tool measures that: for K in --handles, K model
handles, each on its own HIP stream, each running --rounds forwards of --batch sites; one JSON line per K with the
aggregate rate, and a check that every handle's probabilities equal those of the same batch run alone.

usage (GPU box): python tools/bench_small_batches.py --batch 512 --handles 1,2,4,8,16,32
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--handles", default="1,2,4,8,16,32")
    ap.add_argument("--rounds", type=int, default=50)
    ap.add_argument("--check_every_round", action="store_true", help="soak: compare every handle's output of every round "
                    "with the sequential result (slower: a synchronisation per round)")
    args = ap.parse_args(argv)

    import torch
    from deepsignal_plant_amd import synth
    from deepsignal_plant_amd.models import ModelBiLSTM
    assert torch.cuda.is_available(), "needs an MI355X"
    dev = torch.device("cuda", 0)
    B = args.batch
    ks = [int(x) for x in args.handles.split(",")]
    kmax = max(ks)

    def make():
        m = ModelBiLSTM(13, 16, 3, 1, 2, 0, 256, 16, 4, True, True, module="both_bilstm", device=0, init_state="randn",
                        seed=2024)
        m.load_state_dict(synth.random_state_dict(m, seed=1234))
        m.cuda(0).eval()
        m.reserve(B)
        return m
    models = [make() for _ in range(kmax)]
    streams = [torch.cuda.Stream(dev) for _ in range(kmax)]
    batches = [synth.feature_batch(B, device=str(dev), seed=77 + i) for i in range(kmax)]
    # every batch alone, on the default stream: the values a concurrent run has to reproduce
    alone = []
    for i in range(kmax):
        models[0].site_offset = i * B
        alone.append(models[0](*batches[i])[1].clone())
    torch.cuda.synchronize()

    for k in ks:
        outs = [None] * k
        bad_rounds = 0
        for it in range(3 + args.rounds):
            if it == 3:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            for i in range(k):
                with torch.cuda.stream(streams[i]):
                    models[i].site_offset = i * B
                    outs[i] = models[i](*batches[i])[1]
            if args.check_every_round:
                torch.cuda.synchronize()
                bad_rounds += int(not all(bool(torch.equal(outs[i], alone[i])) for i in range(k)))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = all(bool(torch.equal(outs[i], alone[i])) for i in range(k))
        print(json.dumps({"batch": B, "handles": k, "rounds": args.rounds, "sites_per_s": round(k * B * args.rounds / dt, 1),
                          "ms_per_round": round(dt / args.rounds * 1e3, 3), "identical_to_sequential": same,
                          **({"rounds_that_differed": bad_rounds} if args.check_every_round else {})}), flush=True)
        assert bad_rounds == 0
        assert same, "concurrent forwards on separate handles changed the results"
    return 0


if __name__ == "__main__":
    sys.exit(main())
