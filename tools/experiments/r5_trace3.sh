export DSP_AMD_LIB=$PWD/deepsignal_plant_amd/libdsp_amd_trace.so DSP_TWO_STREAMS=0
for b in 512 1024; do
  for h in 0 1; do
    echo "== hand-off $h, comb layer 1"; DSP_LSTM_HANDOFF=$h DSP_TRACE_LAUNCH=3 python3 tools/experiments/r5_trace_cluster.py --batch $b 2>&1 | grep -v amdgpu.ids
  done
done
echo "== hand-off 1, front end (seq), 512"; DSP_LSTM_HANDOFF=1 DSP_TRACE_LAUNCH=0 python3 tools/experiments/r5_trace_cluster.py --batch 512 2>&1 | grep -v amdgpu.ids
