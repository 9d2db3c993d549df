#!/bin/bash
# round-6 measurement set (GPU box), in the order the review asks for: the driver's own commands first,
# then the record -- kernel stats, the six PMC passes with the traffic stamp, per-launch times, the batch sweep, the small-batch
# profiles, the split-precision kernels re-profiled --, then the checks that are new this round (hardware range probe, extents
# sweep under the bounds-recording build, same-box A/B of the descriptors' extents).  Nothing is cut; gpurun_out/r6/ comes back.
#   gpurun --timeout 2400 -- bash tools/r6_final.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6; mkdir -p $O
t0=$(date +%s)
# 1. the GPU suite as the driver runs it (every module in its own child: tests/gpu_isolation.py), uncut
DSP_GPU_SUITE_DIR=$PWD/$O/suite python -m pytest tests -x -q -m gpu --durations=15 -p no:cacheprovider > $O/gputest_full.log 2>&1
echo "GPU suite rc=$? $(( $(date +%s) - t0 )) s: $(grep -v amdgpu.ids $O/gputest_full.log | tail -n 1)"
[ -f $O/suite/deaths.txt ] && cat $O/suite/deaths.txt
# 2. the driver-flag bench line
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags_20steps.json 2> $O/bench_driver_flags.err; echo "bench rc=$?"
# 3. same-box A/B of the descriptors' extents on that line (allocation ends / tight / the 2 GiB windows of rounds 1-5), twice each
for rep in 1 2; do for mode in region wide tight; do
  DSP_RSRC_EXTENTS=$mode python3 bench.py --gpus 1 --steps 20 --warmup 5 --no_cpu_baseline --no_alt 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('extents $mode rep $rep', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done > $O/extents_ab.txt 2>&1
# 4. the record: profiles + PMC passes of both bench configurations, the traffic stamp
bash tools/profile.sh r6 5 > gpurun_out/prof_r6.log 2>&1
bash tools/profile.sh r6_cfg3 5 --model_type seq_bilstm --layernum1 2 > gpurun_out/prof_r6_cfg3.log 2>&1
python3 -c "import json; json.dump([json.load(open(\"gpurun_out/prof_%s/traffic_entry.json\" % d)) for d in (\"r6\", \"r6_cfg3\")], open(\"profiles/traffic.json\", \"w\"), indent=1)"
cp profiles/traffic.json $O/traffic.json
python3 bench.py > $O/bench_default_153steps.json 2> $O/bench_default.err
python3 bench.py --model_type seq_bilstm --layernum1 2 > $O/bench_cfg3_153steps.json 2> $O/bench_cfg3.err
python3 tools/per_launch.py --reps 10 > $O/per_launch_hip_events.txt 2>&1
for b in 512 1024 2048 4096; do echo "== batch $b"; python3 tools/per_launch.py --batch $b --reps 20; done > $O/per_launch_small.log 2>&1
bash tools/batch_sweep.sh $O/batch_sweep.jsonl > $O/batch_sweep.txt 2>&1
bash tools/profile.sh r6_b512 200 --batch 512 > /dev/null 2>&1
bash tools/profile.sh r6_b1024 100 --batch 1024 > /dev/null 2>&1
# 128 sites with and without "x ahead": kernel stats of the recurrent launches, dsp_xahead_kernel next to them
bash tools/profile.sh r6_b128 200 --batch 128 > /dev/null 2>&1
DSP_LSTM_XAHEAD=1 bash tools/profile.sh r6_b128_xahead 200 --batch 128 > /dev/null 2>&1
# the split-precision kernels (two kernel generations since their last profile)
bash tools/profile.sh r6_bf16x9 5 --precision bf16x9 > /dev/null 2>&1
bash tools/profile.sh r6_fp16x3 5 --precision fp16x3 > /dev/null 2>&1
# 4a. all eleven shapes of the extents sweep (the suite runs six): default extents / tight / the bounds-recording build
python3 tools/extents_sweep.py > $O/extents_sweep_region.json 2> $O/extents_sweep_region.err
DSP_RSRC_EXTENTS=tight python3 tools/extents_sweep.py > $O/extents_sweep_tight.json 2> $O/extents_sweep_tight.err
DSP_AMD_LIB=$PWD/deepsignal_plant_amd/libdsp_amd_bounds.so python3 tools/extents_sweep.py > $O/extents_sweep_bounds.json 2> $O/extents_sweep_bounds.err
python3 -c "
import json
r = [json.loads(open('$O/extents_sweep_%s.json' % k).read().strip().splitlines()[-1]) for k in ('region', 'tight', 'bounds')]
print('extents sweep: %d cases; tight == region: %s; bounds build == region: %s' % (len(r[0]), r[1] == r[0], r[2] == r[0]))"
# 4b. "x ahead" (opt-in, never run on a GPU when it was written): bytes and per-forward times with / without it, rings 8 deep too
XAHEAD_AB_REPS=200 python3 tools/xahead_ab.py > $O/xahead_ab.json 2> $O/xahead_ab.err; echo "xahead_ab rc=$?"; tail -c 1500 $O/xahead_ab.json
# 5. the plan's cost model against this box: what dsp_debug_piece_cost says next to what the sweep measured
python3 - <<'PY' > gpurun_out/r6/plan_vs_measured.txt 2>&1
import ctypes, json
from deepsignal_plant_amd import _native as nat
c = nat.ModelCfg(13, 16, 3, 1, 2, 256, 16, 4, 1, 1, 0)
for line in open("gpurun_out/r6/batch_sweep.jsonl"):
    try:
        d = json.loads(line)
    except ValueError:
        continue
    n = d["config"]["batch"]
    if n <= 8192:
        print(n, "measured %.3f ms" % d["ms_per_step"], "model %.3f ms (one piece)" % (nat.lib().dsp_debug_piece_cost(ctypes.byref(c), 256, n) / 1000))
PY
python3 - <<'PY'
import json
for f in ("bench_default_153steps", "bench_cfg3_153steps", "bench_driver_flags_20steps"):
    try:
        d = json.loads(open("gpurun_out/r6/%s.json" % f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable:", e); continue
    r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["frac"], r.get("whole_forward_frac"), r.get("traffic"), r.get("traffic_over_algorithmic"),
          r.get("step_traffic_over_algorithmic"), r.get("avg_launch_ms"), r.get("kernel_src_sha16"), (d.get("cpu_baseline") or {}).get("value"))
PY
cat $O/extents_ab.txt; tail -n 14 $O/batch_sweep.txt; cat $O/plan_vs_measured.txt
echo "total $(( $(date +%s) - t0 )) s"; du -sh gpurun_out | tail -1
