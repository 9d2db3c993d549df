# kernel-trace stats of 200 forwards of 512 (and 1,024) sites with round 4's hand-off and with round 5's: true kernel durations
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r5/kstats; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for b in 512 1024; do
 for h in 0 1; do
  RAW=/tmp/ks_${b}_$h; rm -rf $RAW
  DSP_LSTM_HANDOFF=$h rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o k -- python3 $REPO/bench.py --steps 200 --warmup 5 --no_cpu_baseline --no_alt --batch $b > $OUT/bench_${b}_h$h.log 2>&1
  find $RAW -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_${b}_h$h.csv \;
  python3 - $OUT/kernel_stats_${b}_h$h.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(sys.argv[1].split("/")[-1])
for r in rows:
    if "dsp_" in r["Name"]:
        print("  %-58s calls %5s avg %9.1f us" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  tail -1 $OUT/bench_${b}_h$h.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  bench ms_per_step', d['ms_per_step'], 'events_off', d['roofline']['ms_per_step_events_off'])"
  rm -rf $RAW
 done
done
