#!/bin/bash
# round 5 regression check of the CLI pipeline at the round's last build: 4 M rows (8.35 GB of text) end to end with both parsers, the same
# under DSP_SLOT_CANARY=1 (what the debugging mode costs), and 100,000 rows (BASELINE configs[0]) -- one GPU
export DSP_WORK=/tmp/dsp_pipe DSP_BENCH_NO_DSPF=1
out=gpurun_out/r5; mkdir -p $out
: > $out/pipeline_cli.jsonl
for mode in device host; do
  DSP_PARSE_ON=$mode DSP_BENCH_THREADS=1,4 python tools/bench_pipeline.py 4000000 2>/dev/null | grep '^{' >> $out/pipeline_cli.jsonl
done
DSP_SLOT_CANARY=1 DSP_PARSE_ON=device DSP_BENCH_THREADS=4 python tools/bench_pipeline.py 4000000 2>/dev/null | grep '^{' | sed 's/^{/{"canary": 1, /' >> $out/pipeline_cli.jsonl
DSP_PARSE_ON=device DSP_BENCH_THREADS=4 python tools/bench_pipeline.py 100000 2>/dev/null | grep '^{' >> $out/pipeline_cli.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r5/pipeline_cli.jsonl"):
    d = json.loads(l)
    print("rows %8d parse_on %-6s -p %2d%s: call_mods %.2f s (process %.2f s)  %.3f M sites/s" % (d["rows"], d.get("parse_on"), d["parse_threads"], " CANARY" if d.get("canary") else "", d["call_mods_s"], d["process_wall_s"], d["sites_per_s"] / 1e6))
PY
