#!/usr/bin/env python3
"""bench.py -- methylation sites/s of the call_mods forward (both_bilstm bn13_sn16) on N x MI355X.

A "step" is one pass of the hot path over one batch of 65,536 synthetic sites whose feature tensors are
already resident in HBM (BASELINE.json configs[1]: 10M synthetic sites, fp32, batch 65536; 153 steps =
10,027,008 sites).  Sites are independent, so N GPUs range-shard the site index space with no data-path
collective ("weak" scaling: every rank runs K steps of its own range).  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 65536
FP32_MATRIX_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
CPU_BASELINE_TARGET_S = float(os.environ.get("DSP_CPU_BASELINE_S", 15.0))  # bounded sample: ~15 s of host work


def cpu_baseline(model_cfg_kwargs, sd_numpy, seed):
    """The oracle's C port (oracle/dsp_oracle.c, OpenMP over site blocks) timed on this box's host cores on
    a bounded sample of the same workload (same weights, same synthetic row statistics, batch semantics are
    irrelevant on the CPU: sites are independent)."""
    import numpy as np
    from oracle import c_oracle as oc
    from oracle import forward_np as onp
    cfg = onp.OracleConfig(**model_cfg_kwargs)
    threads = oc.num_threads()
    probe = 32 * max(1, threads)
    ins = onp.make_inputs(cfg, probe, 7)
    t0 = time.time()
    oc.forward(cfg, sd_numpy, *ins, init_mode="philox", seed=seed, nthreads=threads)
    rate = probe / max(time.time() - t0, 1e-6)
    n = int(max(probe, min(rate * CPU_BASELINE_TARGET_S, 1 << 20)))
    n = (n + 15) // 16 * 16
    ins = onp.make_inputs(cfg, n, 8)
    t0 = time.time()
    oc.forward(cfg, sd_numpy, *ins, init_mode="philox", seed=seed, nthreads=threads)
    dt = time.time() - t0
    return {"value": round(n / dt, 1), "unit": "sites/s", "cores": threads, "kind": "port",
            "sample": "%d synthetic sites (same model/weights/row statistics), oracle/dsp_oracle.c fp32 + OpenMP, %.1f s"
                      % (n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=153)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--model_type", default="both_bilstm")
    ap.add_argument("--layernum1", type=int, default=3)
    ap.add_argument("--hid_rnn", type=int, default=256)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--precision", default=os.environ.get("DSP_PRECISION", "fp32"), choices=["fp32", "bf16x6", "bf16x9", "fp16x3"],
                    help="products of the combined stack: fp32 MFMA (default, what `value` is measured in) or the opt-in "
                         "split-bf16 emulation (include/dsp_amd.h DSP_PREC_*)")
    ap.add_argument("--no_alt", action="store_true", help="skip the extra fp16x3 measurement reported under alt_precision")
    ap.add_argument("--gather", action="store_true", help="optional final RCCL all_gather of per-site probs")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from deepsignal_plant_amd import synth
    from deepsignal_plant_amd.models import ModelBiLSTM

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    assert ndev > 0, "bench.py needs an MI355X"
    dev_index = local_rank % ndev
    # one process per GPU over RCCL; if fewer GPUs than ranks are visible (the 1-GPU dev box), the ranks share
    # GPU 0 and the two control-plane collectives (barrier, max of the wall time) run over gloo instead
    backend = "nccl" if ndev >= world else "gloo"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    local_rank = dev_index

    K, W, B = args.steps, args.warmup, args.batch
    model = ModelBiLSTM(13, 16, args.layernum1, 1, 2, 0, args.hid_rnn, 16, 4, True, True, module=args.model_type,
                        device=local_rank, init_state="randn", seed=2024)
    sd = synth.random_state_dict(model, seed=1234)
    model.load_state_dict(sd)
    model.cuda(local_rank).eval()
    model.set_precision(args.precision)
    model.reserve(B)
    flops_site = model.flops_per_site()

    # resident synthetic input: distinct batches (up to 160 = 10.9 GB); more steps cycle over them
    nb = min(K, 160) if K > 0 else 1
    batches = [synth.feature_batch(B, device=str(dev), seed=1000 * rank + i) for i in range(nb)]
    site0 = rank * K * B  # this rank's range of the global site index space
    outs = None

    def step(i):
        model.site_offset = site0 + i * B
        return model(*batches[i % nb])

    for i in range(W):
        step(i)
    torch.cuda.synchronize()
    model.profile(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        outs = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = model.profile_read()
    model.profile(False)

    if world > 1:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if args.gather:  # optional final gather of the per-site probabilities of the last step (RCCL over xGMI)
            src = outs[1] if backend == "nccl" else outs[1].cpu()
            gathered = [torch.empty_like(src) for _ in range(world)]
            dist.all_gather(gathered, src)
    assert outs is not None and bool(torch.isfinite(outs[1]).all())

    # opt-in mode, reported next to the headline (never as `value`): the same steps with the LSTMs' fp32 products
    # emulated by split low-precision pieces on the fast matrix pipes (include/dsp_amd.h DSP_PREC_FP16X3), and how far
    # its probabilities are from the fp32 path's
    alt = None
    if world == 1 and args.precision == "fp32" and not args.no_alt and K > 0:
        ka = min(K, 40)
        model.site_offset = site0
        ref = model(*batches[0])[1].clone()
        model.set_precision("fp16x3")
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
        model.set_precision("fp32")
        alt = {"dtype": "f32 via fp16x3 (2 fp16 pieces per operand, 3 piece products, f32 accumulate; front ends bf16x6)",
               "value": round(ka * B / ta, 1), "unit": "sites/s", "steps": ka, "ms_per_step": round(ta / ka * 1e3, 3),
               "max_abs_dprob_vs_fp32_path": dmax, "how": "python bench.py --precision fp16x3"}

    if rank == 0:
        total_sites = world * K * B
        value = total_sites / dt
        # roofline of the dominant kernel, dsp_lstm4_kernel<false> = the launches of the combined BiLSTM stack (the
        # front-end launches run the <true> instantiation and are listed separately by rocprofv3): algorithmic
        # FLOPs of those launches / their summed duration, durations from HIP events on the launch stream
        comb_ms = [ms for name, ms in prof if name == "lstm_comb"]
        all_ms = sum(ms for _, ms in prof)
        H, T = model.hidden_size, model.seq_len
        comb_flops_site = sum(2 * (2 * T * 4 * H * ((H if k == 0 else 2 * H) + H)) for k in range(model.num_layers1))
        n_lstm = max(len(comb_ms), 1)
        avg_ms = sum(comb_ms) / n_lstm
        flops_per_launch = comb_flops_site * B * K / n_lstm
        nprod = {"fp32": 1, "bf16x6": 6, "bf16x9": 9, "fp16x3": 3}[args.precision]
        # split-bf16 modes execute nprod bf16 piece products per fp32 product: price those against the bf16 peak
        peak = FP32_MATRIX_PEAK_TFLOPS if nprod == 1 else 16 * FP32_MATRIX_PEAK_TFLOPS
        achieved = nprod * flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        lstm_ms = comb_ms
        traffic = None
        try:  # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/)
            tj = json.load(open(os.path.join(ROOT, "profiles", "r1", "traffic.json")))
            if B == BATCH and args.model_type == "both_bilstm" and args.layernum1 == 3 and args.hid_rnn == 256:
                traffic = tj["hbm_bytes_per_launch"]
        except Exception:
            pass
        line = {
            "metric": "methylation sites/sec, %s bn13_sn16" % args.model_type, "value": round(value, 1), "unit": "sites/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / max(K, 1) * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if nprod == 1 else "f32 via %s (%s pieces per operand, %d piece products, f32 accumulate)" % (
                args.precision, "2 fp16" if nprod == 3 else "3 bf16", nprod),
            "data": "synthetic",
            "config": {"workload": "%d synthetic sites, %s bn13_sn16 fp32, batch %d on %dxMI355X (BASELINE.json configs[%d])"
                                   % (total_sites, args.model_type, B, world,
                                      2 if args.model_type == "seq_bilstm" else (1 if world == 1 else 3)),
                       "batch": B, "sites": total_sites, "layernum1": args.layernum1, "hid_rnn": args.hid_rnn,
                       "init_state": "in-kernel Philox N(0,1) (stand-in for torch.randn, models.py:169-176)",
                       "weights": "seeded random state_dict, PyTorch default-init scale", "parallelism": "range-shard x%d" % world,
                       "flops_per_site": flops_site},
            "roofline": {"bound": "mfma", "kernel": "dsp_lstm4_kernel<false>" if nprod == 1 else "dsp_lstm6_kernel<%d>" % nprod,
                         "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "traffic": traffic if nprod == 1 else None, "avg_launch_ms": round(avg_ms, 4), "launches": len(lstm_ms),
                         "flops_per_launch": flops_per_launch,
                         "whole_forward_tflops": round(value / world * flops_site / 1e12, 2),
                         "hbm_gbps_algorithmic": round(value / world * 1048 / 1e9, 3),
                         "kernel_time_frac_of_wall": round(all_ms * 1e-3 / dt, 4)},
        }
        if alt is not None:
            line["alt_precision"] = alt
        if not args.no_cpu_baseline and world == 1:
            try:
                kw = dict(num_layers1=args.layernum1, hidden_size=args.hid_rnn, module=args.model_type)
                line["cpu_baseline"] = cpu_baseline(kw, {k: v.numpy() for k, v in sd.items()}, 2024)
            except Exception as e:  # the baseline must never take the GPU number down with it
                line["cpu_baseline"] = {"value": None, "unit": "sites/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
