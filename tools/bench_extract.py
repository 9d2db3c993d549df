#!/usr/bin/env python3
"""Throughput of the extraction stage on this box: synthetic reads of realistic size (about 9 samples per base)
-> device features.  Prints one JSON line per normalisation method with per-kernel HIP-event times, the
algorithmic bytes they move, and the oracle (numpy restatement of the reference) timed on a bounded sample."""
import json
import os
import sys
import time


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from deepsignal_plant_amd import extract_features as ef
    from deepsignal_plant_amd import reads as R
    from oracle import extract_np as ox
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    mean_bases = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    rs = R.synth_reads(n_reads, seed=1, mean_bases=mean_bases)
    samples = sum(len(r.raw) for r in rs)
    bases = sum(len(r.ev_len) for r in rs)
    for method in ("mad", "zscore"):
        fx = ef.FeatureExtractor(normalize_method=method, seed=1)
        out = fx.extract(rs)  # warm-up (allocator, code objects)
        torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fx.extract(rs)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        sites = out.n
        # the same batches through the staged pipeline: host halves of later batches under the GPU halves of earlier ones
        piped = {}
        for workers in (1, 4, 8):
            nb = 12
            for o in fx.extract_stream(((rs, None) for _ in range(3)), workers=workers):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for o in fx.extract_stream(((rs, None) for _ in range(nb)), workers=workers):
                assert o.n == sites
            torch.cuda.synchronize()
            piped[workers] = (time.perf_counter() - t0) / nb
        # GPU time of one batch alone (events around the launch half)
        sg = fx.stage(rs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        fx.launch(sg)
        e1.record()
        torch.cuda.synchronize()
        gpu_ms = e0.elapsed_time(e1)
        # algorithmic HBM bytes: raw read by each of the three kernels' first touch (2 B/sample), events 17 B/base,
        # per-base stats 28 B/base written + read by the gather, features 1,025 B/site written
        L, S = fx.L, fx.S
        alg_bytes = samples * 2 + bases * 17 + bases * 28 + sites * (L * (1 + 4 + 4 + 4) + L * S * 4)
        t0 = time.perf_counter()
        k = max(1, min(len(rs), 8))
        feats = ox.extract_features(rs[:k], method, ["CG"], 0, None, 13, 16, 1, sampler="hash", seed=1)
        cpu = time.perf_counter() - t0
        print(json.dumps({"stage": "extract", "normalize": method, "reads": n_reads, "samples": samples, "bases": bases,
                          "sites": sites, "wall_ms_per_batch": round(wall * 1e3, 2),
                          "gpu_ms_per_batch_uploads_plus_kernels": round(gpu_ms, 2),
                          "pipelined_wall_ms_per_batch": {str(k): round(v * 1e3, 2) for k, v in piped.items()},
                          "pipelined_sites_per_s_8_workers": round(sites / piped[8], 1),
                          "sites_per_s": round(sites / wall, 1), "msamples_per_s": round(samples / wall / 1e6, 1),
                          "algorithmic_gb": round(alg_bytes / 1e9, 4), "algorithmic_gbps_wall": round(alg_bytes / wall / 1e9, 1),
                          "cpu_oracle_sites_per_s": round(len(feats) / cpu, 1), "cpu_oracle_reads": k}), flush=True)


if __name__ == "__main__":
    main()
