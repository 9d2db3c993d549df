cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for b in 4096 2048 512; do
    echo "== batch $b"; python tools/per_launch.py --batch $b --reps 30 2>/dev/null | grep "lstm_\|sum"
  done
done
timeout 300 python tools/experiments/r4_cluster_check.py 2>&1 | tail -14
