#!/bin/bash
# round-5 measurement set (GPU box): profiles + PMC passes of both bench configurations (with the traffic stamp), the bench lines,
# per-launch times, the batch-size sweep, the small-batch profiles, kernel stats and soaks of the round-5 small-batch forms
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r5; mkdir -p $O
bash tools/profile.sh r5 5 > gpurun_out/prof_r5.log 2>&1
bash tools/profile.sh r5_cfg3 5 --model_type seq_bilstm --layernum1 2 > gpurun_out/prof_r5_cfg3.log 2>&1
python3 -c "import json; json.dump([json.load(open(\"gpurun_out/prof_%s/traffic_entry.json\" % d)) for d in (\"r5\", \"r5_cfg3\")], open(\"profiles/traffic.json\", \"w\"), indent=1)"
cp profiles/traffic.json $O/traffic.json
python3 bench.py > $O/bench_default_153steps.json 2> $O/bench_default.err
python3 bench.py --model_type seq_bilstm --layernum1 2 > $O/bench_cfg3_153steps.json 2> $O/bench_cfg3.err
python3 bench.py --steps 20 --warmup 3 > $O/bench_driver_flags_20steps.json 2>/dev/null
python3 tools/per_launch.py --reps 10 > $O/per_launch_hip_events.txt 2>&1
for b in 512 1024 2048 4096; do echo "== batch $b"; python3 tools/per_launch.py --batch $b --reps 20; done > $O/per_launch_small.log 2>&1
bash tools/batch_sweep.sh $O/batch_sweep.jsonl > $O/batch_sweep.txt 2>&1
bash tools/profile.sh r5_b512 200 --batch 512 > /dev/null 2>&1
bash tools/profile.sh r5_b1024 100 --batch 1024 > /dev/null 2>&1
bash tools/experiments/r5_small_ab.sh 512,1024,2048,4096 300 > $O/small_batch_ab_final.txt 2>&1
python3 tools/experiments/r4_soak_small.py 1000 > $O/soak_small_batches.txt 2>&1
for spec in "512 1,2,4,8,16,32" "2048 1,2,4,8"; do set -- $spec
  python3 tools/bench_small_batches.py --batch $1 --handles $2 --rounds 40 2>/dev/null
done > $O/small_batches.jsonl
python3 - <<'PY'
import json
for f in ("bench_default_153steps", "bench_cfg3_153steps", "bench_driver_flags_20steps"):
    d = json.loads(open("gpurun_out/r5/%s.json" % f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["frac"], r.get("whole_forward_frac"), r.get("traffic"), r.get("traffic_over_algorithmic"),
          r.get("step_traffic_over_algorithmic"), r.get("avg_launch_ms"), r.get("kernel_src_sha16"), (d.get("cpu_baseline") or {}).get("value"))
PY
tail -n 14 $O/batch_sweep.txt
grep -v amdgpu $O/small_batch_ab_final.txt | head -8
tail -2 $O/soak_small_batches.txt
