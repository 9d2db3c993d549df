#!/bin/bash
# round-2 measurement set (GPU box): profiles of both bench configurations, the bench lines themselves, the CLI pipelines
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r2
bash tools/profile.sh r2 5 > gpurun_out/prof_r2.log 2>&1
bash tools/profile.sh r2_cfg3 5 --model_type seq_bilstm --layernum1 2 > gpurun_out/prof_r2_cfg3.log 2>&1
python3 bench.py > gpurun_out/r2/bench_default.json 2> gpurun_out/r2/bench_default.err
python3 bench.py --model_type seq_bilstm --layernum1 2 > gpurun_out/r2/bench_cfg3.json 2> gpurun_out/r2/bench_cfg3.err
python3 bench.py --gpus 2 --steps 20 --no_cpu_baseline > gpurun_out/r2/bench_2ranks_shared_gpu.json 2> gpurun_out/r2/bench_2ranks.err
python3 tools/per_launch.py --reps 10 > gpurun_out/r2/per_launch_hip_events.txt 2>&1
python3 tools/per_launch.py --reps 10 --model_type seq_bilstm --layernum1 2 > gpurun_out/r2/per_launch_hip_events_cfg3.txt 2>&1
python3 tools/bench_pipeline.py 100000 4000000 > gpurun_out/r2/pipeline_cli.jsonl 2> gpurun_out/r2/pipeline_cli.err
python3 tools/bench_reads_pipeline.py > gpurun_out/r2/pipeline_reads.jsonl 2> gpurun_out/r2/pipeline_reads.err
python3 tools/bench_freq.py > gpurun_out/r2/freq_host.txt 2>&1
tail -n 2 gpurun_out/r2/*.json gpurun_out/r2/*.jsonl | cut -c1-400
# micro-benchmarks behind DESIGN.md section 3 (MFMA / VALU overlap, cell-phase cost)
for m in mfma_cell_overlap mfma_same_wave cell_phase trans_rate; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/$m tools/micro/$m.hip 2>/dev/null && timeout 200 /tmp/$m > gpurun_out/r2/micro_$m.txt 2>&1
done
export DSP_AMD_LIB=deepsignal_plant_amd/libdsp_amd_trace.so
{ for LW in "0 0" "1 0" "3 0" "3 4"; do set -- $LW; L=$1; W=$2; echo "== launch $L (0 lstm_seq, 1 lstm_signal, 3 combined layer 1), stamping wave $W"; DSP_TRACE_LAUNCH=$L DSP_TRACE_WAVE=$W timeout 200 python3 tools/trace_lstm.py --cus 1; done; } > gpurun_out/r2/lstm_wave_trace_s_memtime.txt 2>&1
unset DSP_AMD_LIB
python3 tools/bench_fast5_pipeline.py > gpurun_out/r2/pipeline_fast5.jsonl 2> gpurun_out/r2/pipeline_fast5.err
python3 tools/parity_sweep.py 300 > gpurun_out/r2/parity_sweep.txt 2>&1
