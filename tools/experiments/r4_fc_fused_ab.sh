cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
(python -m pytest tests/test_gpu_parity.py tests/test_gpu_windows.py -q -x 2>&1 | tail -3
for i in 1 2 3; do for f in 0 1; do echo "== DSP_FC_FUSED=$f"; DSP_FC_FUSED=$f python tools/per_launch.py --reps 20 2>/dev/null | grep "fc_\|sum"; done; done
for f in 0 1; do echo "== DSP_FC_FUSED=$f batch 4096"; DSP_FC_FUSED=$f python tools/per_launch.py --batch 4096 --reps 30 2>/dev/null | grep "fc_\|sum"; done
) > gpurun_out/r4/fc_fused_ab.txt 2>&1
cat gpurun_out/r4/fc_fused_ab.txt
