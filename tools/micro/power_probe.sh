for p in fp32 bf16x6; do
  DSP_PRECISION=$p python bench.py --steps 300 --warmup 3 --no_cpu_baseline > /tmp/b_$p.json 2>/dev/null &
  PID=$!
  sleep 9
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Graphics Package Power" | tr '\n' ' '; echo; sleep 2; done
  wait $PID
  python3 -c "import json;d=json.loads(open('/tmp/b_$p.json').read().strip().splitlines()[-1]);print('$p', d['value'])"
done
