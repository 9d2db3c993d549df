cd $GRAFT_REPO_ROOT
for rb in 4 8; do echo "RB=$rb"; DSP_PARSE_RB=$rb timeout 200 python tools/experiments/r4_parse_kernel_time.py 2>&1 | tail -1; done
