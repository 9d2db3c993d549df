#!/bin/bash
# round-3 measurement set (GPU box): profiles + PMC passes of both bench configurations (with the traffic stamp), the
# bench lines, per-launch times, clock / power of fp32 vs bf16x9, the CLI pipelines (plain, BGZF, foreign .gz, --freq_file)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r3
bash tools/profile.sh r3 5 > gpurun_out/prof_r3.log 2>&1
bash tools/profile.sh r3_cfg3 5 --model_type seq_bilstm --layernum1 2 > gpurun_out/prof_r3_cfg3.log 2>&1
# the bench lines below quote the traffic of THIS build: put the fresh entries where bench.py looks for them
python3 -c "import json; json.dump([json.load(open(\"gpurun_out/prof_%s/traffic_entry.json\" % d)) for d in (\"r3\", \"r3_cfg3\")], open(\"profiles/traffic.json\", \"w\"), indent=1)"
python3 bench.py > gpurun_out/r3/bench_default_153steps.json 2> gpurun_out/r3/bench_default.err
python3 bench.py --model_type seq_bilstm --layernum1 2 > gpurun_out/r3/bench_cfg3_153steps.json 2> gpurun_out/r3/bench_cfg3.err
python3 bench.py --steps 191 --no_cpu_baseline --no_alt > gpurun_out/r3/bench_config4_share_of_one_gpu_191steps.json 2>/dev/null
python3 tools/per_launch.py --reps 10 > gpurun_out/r3/per_launch_hip_events.txt 2>&1
python3 tools/per_launch.py --reps 10 --model_type seq_bilstm --layernum1 2 > gpurun_out/r3/per_launch_hip_events_cfg3.txt 2>&1
bash tools/micro/power_probe.sh fp32 bf16x9 > gpurun_out/r3/power_probe.txt 2>&1
python3 tools/bench_pipeline_gz.py 4000000 > gpurun_out/r3/pipeline_gz.jsonl 2> gpurun_out/r3/pipeline_gz.err
python3 tools/bench_pipeline_freq.py 4000000 > gpurun_out/r3/pipeline_freq.jsonl 2> gpurun_out/r3/pipeline_freq.err
tail -n 3 gpurun_out/r3/*.json gpurun_out/r3/*.jsonl gpurun_out/r3/power_probe.txt | cut -c1-500
