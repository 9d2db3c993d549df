#!/bin/bash
# After a change to the code of dsp_kernels.hip / dsp_kernels.h / dsp_capi.cpp (GPU box): the full -m gpu suite, both
# profiles with their PMC passes and the traffic stamp, the two committed bench lines, per-launch times, the batch-size
# sweep and the concurrent small batches.  Results under gpurun_out/ (copy into profiles/ what is to be kept).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3
python3 -m pytest tests -q -m gpu -s > gpurun_out/r3/gputest_full.log 2>&1; tail -n 3 gpurun_out/r3/gputest_full.log
bash tools/profile.sh r3 5 > gpurun_out/prof_r3.log 2>&1
bash tools/profile.sh r3_cfg3 5 --model_type seq_bilstm --layernum1 2 > gpurun_out/prof_r3_cfg3.log 2>&1
python3 -c "import json; json.dump([json.load(open(\"gpurun_out/prof_%s/traffic_entry.json\" % d)) for d in (\"r3\", \"r3_cfg3\")], open(\"profiles/traffic.json\", \"w\"), indent=1)"
python3 bench.py > gpurun_out/r3/bench_default_153steps.json 2> gpurun_out/r3/bench_default.err
python3 bench.py --model_type seq_bilstm --layernum1 2 > gpurun_out/r3/bench_cfg3_153steps.json 2> gpurun_out/r3/bench_cfg3.err
python3 tools/per_launch.py --reps 10 > gpurun_out/r3/per_launch_hip_events.txt 2>&1
bash tools/batch_sweep.sh gpurun_out/r3/batch_sweep.jsonl > /dev/null 2>&1
for q in 4 16 32; do for spec in "512 1,2,4,8,16,32" "2048 1,2,4,8"; do set -- $spec
  GPU_MAX_HW_QUEUES=$q python3 tools/bench_small_batches.py --batch $1 --handles $2 --rounds 40 2>/dev/null | sed "s/^{/{\"GPU_MAX_HW_QUEUES\": $q, /"
done; done > gpurun_out/r3/small_batches.jsonl
python3 - <<'PY'
import json
for f in ("bench_default_153steps", "bench_cfg3_153steps"):
    d = json.loads(open("gpurun_out/r3/%s.json" % f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["frac"], r.get("whole_forward_frac"), r.get("traffic"), r.get("traffic_over_algorithmic"),
          r.get("step_traffic_over_algorithmic"), r.get("avg_launch_ms"), r.get("kernel_src_sha16"))
PY
grep -h "dsp_lstm_kernel<0, 1, 0>\|dsp_lstm_kernel<0,1,0>" gpurun_out/prof_r3/kernel_stats.csv | head -2
tail -n 12 gpurun_out/r3/batch_sweep.jsonl | cut -c1-200 | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'].get('batch'), d['value'], d['ms_per_step'])" 2>/dev/null
cut -c1-130 gpurun_out/r3/small_batches.jsonl
