#!/bin/bash
# round-4 measurement set (GPU box): profiles + PMC passes of both bench configurations (with the traffic stamp) and of the two
# opt-in split-precision modes the bench line quotes under alt_precision, the bench lines, per-launch times, clock / power by
# precision, the batch-size sweep, concurrent small batches
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4
bash tools/profile.sh r4 5 > gpurun_out/prof_r4.log 2>&1
bash tools/profile.sh r4_cfg3 5 --model_type seq_bilstm --layernum1 2 > gpurun_out/prof_r4_cfg3.log 2>&1
bash tools/profile.sh r4_fp16x3 5 --precision fp16x3 > gpurun_out/prof_r4_fp16x3.log 2>&1
bash tools/profile.sh r4_bf16x9 5 --precision bf16x9 > gpurun_out/prof_r4_bf16x9.log 2>&1
# the bench lines below quote the traffic of THIS build: put the fresh entries where bench.py looks for them
python3 -c "import json; json.dump([json.load(open(\"gpurun_out/prof_%s/traffic_entry.json\" % d)) for d in (\"r4\", \"r4_cfg3\")], open(\"profiles/traffic.json\", \"w\"), indent=1)"
cp profiles/traffic.json gpurun_out/r4/traffic.json
python3 bench.py > gpurun_out/r4/bench_default_153steps.json 2> gpurun_out/r4/bench_default.err
python3 bench.py --model_type seq_bilstm --layernum1 2 > gpurun_out/r4/bench_cfg3_153steps.json 2> gpurun_out/r4/bench_cfg3.err
python3 bench.py --steps 20 --warmup 3 > gpurun_out/r4/bench_driver_flags_20steps.json 2>/dev/null
python3 bench.py --steps 191 --no_cpu_baseline --no_alt > gpurun_out/r4/bench_config4_share_of_one_gpu_191steps.json 2>/dev/null
python3 tools/per_launch.py --reps 10 > gpurun_out/r4/per_launch_hip_events.txt 2>&1
python3 tools/per_launch.py --reps 10 --model_type seq_bilstm --layernum1 2 > gpurun_out/r4/per_launch_hip_events_cfg3.txt 2>&1
for b in 512 1024 2048 4096; do echo "== batch $b"; python3 tools/per_launch.py --batch $b --reps 20; done > gpurun_out/r4/per_launch_small.log 2>&1
bash tools/micro/power_probe.sh fp32 bf16x9 fp16x3 > gpurun_out/r4/power_probe.txt 2>&1
bash tools/batch_sweep.sh gpurun_out/r4/batch_sweep.jsonl > gpurun_out/r4/batch_sweep.txt 2>&1
for spec in "512 1,2,4,8,16,32" "2048 1,2,4,8"; do set -- $spec
  python3 tools/bench_small_batches.py --batch $1 --handles $2 --rounds 40 2>/dev/null
done > gpurun_out/r4/small_batches.jsonl
python3 - <<'PY'
import json
for f in ("bench_default_153steps", "bench_cfg3_153steps", "bench_driver_flags_20steps"):
    d = json.loads(open("gpurun_out/r4/%s.json" % f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["frac"], r.get("whole_forward_frac"), r.get("traffic"), r.get("traffic_over_algorithmic"),
          r.get("step_traffic_over_algorithmic"), r.get("avg_launch_ms"), r.get("kernel_src_sha16"), (d.get("cpu_baseline") or {}).get("value"))
PY
tail -n 14 gpurun_out/r4/batch_sweep.txt
tail -n 4 gpurun_out/r4/power_probe.txt | cut -c1-300
