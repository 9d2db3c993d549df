export DSP_AMD_LIB=$PWD/deepsignal_plant_amd/libdsp_amd_trace.so DSP_LSTM_PERSIST=0 DSP_TWO_STREAMS=0
for b in 512 1024; do
  for l in 0 2 3; do
    DSP_TRACE_LAUNCH=$l python3 tools/experiments/r5_trace_cluster.py --batch $b 2>&1 | grep -v amdgpu.ids
  done
done
