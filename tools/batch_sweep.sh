#!/bin/bash
# sites/s of the forward against the batch size handed to dsp_forward (the reference's CLI default is 512,
# call_modifications.py:147; BASELINE.json's configs 2-4 use 65,536).  One bench.py line per batch size.
# usage (GPU box): bash tools/batch_sweep.sh gpurun_out/r3/batch_sweep.jsonl
out=${1:-gpurun_out/batch_sweep.jsonl}
mkdir -p "$(dirname "$out")"
: > "$out"
for b in 64 512 1024 1100 2048 3000 4096 5000 8192 10000 16384 32768 65536 66000 131072 262144; do
    steps=$(( 6000000 / b )); [ $steps -gt 2000 ] && steps=2000; [ $steps -lt 20 ] && steps=20
    python bench.py --batch $b --steps $steps --warmup 5 --no_cpu_baseline --no_alt | tail -1 >> "$out"
done
python - "$out" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line)
    print(d["config"].get("batch", "?"), d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("whole_forward_frac"))
PY
