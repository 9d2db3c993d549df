cd $GRAFT_REPO_ROOT
for e in "X=1" "X=2"; do
  echo "== $e"
  env $e DSP_PARSE_DEBUG=1 python tools/experiments/r4_parse_diff.py 120000 2>&1 | tail -8 | cut -c1-250
done
