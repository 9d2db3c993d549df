#!/bin/bash
# clock and package power while bench.py runs in a given precision mode (rocm-smi sampled during the timed region)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for p in ${*:-fp32 bf16x9 fp16x3}; do
  python bench.py --precision $p --steps 300 --warmup 3 --no_cpu_baseline --no_alt > /tmp/b_$p.json 2>/dev/null &
  PID=$!
  sleep 12
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Graphics Package Power|Average Graphics" | tr '\n' ' '; echo; sleep 2; done
  wait $PID
  python3 -c "import json;d=json.loads(open('/tmp/b_$p.json').read().strip().splitlines()[-1]);print('$p', d['value'], d['roofline']['ms_per_step_by_launch'])"
done
