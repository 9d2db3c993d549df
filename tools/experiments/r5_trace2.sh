export DSP_AMD_LIB=$PWD/deepsignal_plant_amd/libdsp_amd_trace.so DSP_TWO_STREAMS=0
for b in 512 2048; do
  echo "== one launch per layer, layer 2"; DSP_LSTM_PERSIST=0 DSP_TRACE_LAUNCH=4 python3 tools/experiments/r5_trace_cluster.py --batch $b 2>&1 | grep -v amdgpu.ids
  echo "== persistent stack, layer 2";     DSP_LSTM_PERSIST=1 DSP_TRACE_LAUNCH=4 python3 tools/experiments/r5_trace_cluster.py --batch $b 2>&1 | grep -v amdgpu.ids
done
