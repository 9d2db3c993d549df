#!/bin/bash
# after a change to the code of dsp_kernels.hip / dsp_kernels.h / dsp_capi.cpp (GPU box): both profiles with their PMC passes and
# the traffic stamp, the committed bench lines, the batch-size sweep, the small-batch profiles
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
bash tools/profile.sh r4 5 > gpurun_out/prof_r4.log 2>&1
bash tools/profile.sh r4_cfg3 5 --model_type seq_bilstm --layernum1 2 > gpurun_out/prof_r4_cfg3.log 2>&1
python3 -c "import json; json.dump([json.load(open(\"gpurun_out/prof_%s/traffic_entry.json\" % d)) for d in (\"r4\", \"r4_cfg3\")], open(\"profiles/traffic.json\", \"w\"), indent=1)"
cp profiles/traffic.json gpurun_out/r4/traffic.json
python3 bench.py > gpurun_out/r4/bench_default_153steps.json 2> gpurun_out/r4/bench_default.err
python3 bench.py --model_type seq_bilstm --layernum1 2 > gpurun_out/r4/bench_cfg3_153steps.json 2> gpurun_out/r4/bench_cfg3.err
python3 bench.py --steps 20 --warmup 3 > gpurun_out/r4/bench_driver_flags_20steps.json 2>/dev/null
bash tools/batch_sweep.sh gpurun_out/r4/batch_sweep.jsonl > gpurun_out/r4/batch_sweep.txt 2>&1
for b in 512 1024 2048 4096; do echo "== batch $b"; python3 tools/per_launch.py --batch $b --reps 20; done > gpurun_out/r4/per_launch_small.log 2>&1
bash tools/profile.sh r4_b512 200 --batch 512 > /dev/null 2>&1
bash tools/profile.sh r4_b4096 50 --batch 4096 > /dev/null 2>&1
python3 - <<'PY'
import json
for f in ("bench_default_153steps", "bench_cfg3_153steps", "bench_driver_flags_20steps"):
    d = json.loads(open("gpurun_out/r4/%s.json" % f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["frac"], r.get("whole_forward_frac"), r.get("traffic"), r.get("traffic_over_algorithmic"),
          r.get("step_traffic_over_algorithmic"), r.get("avg_launch_ms"), r.get("kernel_src_sha16"), (d.get("cpu_baseline") or {}).get("value"))
PY
tail -n 12 gpurun_out/r4/batch_sweep.txt
grep -h "dsp_lstm_kernel<0, 1, 0>" gpurun_out/prof_r4/kernel_stats.csv | head -2
