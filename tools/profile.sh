#!/bin/bash
# Profile recipe (GPU box): kernel trace + stats, then separate PMC passes (never combined with tracing
# domains).  Small summaries are written under gpurun_out/prof_$TAG; raw traces are deleted on the box.
# usage: tools/profile.sh TAG [STEPS] [extra bench.py flags, e.g. --model_type seq_bilstm --layernum1 2]
set -u
TAG=${1:-r2}
STEPS=${2:-5}
shift; shift
EXTRA="$*"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
RAW=/tmp/prof_raw_$TAG
rm -rf $OUT $RAW; mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
echo "python3 bench.py --steps $STEPS --warmup 1 --no_cpu_baseline --no_alt $EXTRA" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o bench -- python3 $REPO/bench.py --steps $STEPS --warmup 1 --no_cpu_baseline --no_alt $EXTRA > $OUT/bench_trace.log 2>&1
find $RAW/trace -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
python3 $REPO/tools/summarize_prof.py trace $RAW/trace $OUT/kernel_trace_summary.txt
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $RAW/pmc_$N -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_alt $EXTRA > $OUT/pmc_$N.log 2>&1
  python3 $REPO/tools/summarize_prof.py pmc $RAW/pmc_$N $OUT/pmc_$N.txt
done
# HBM-side bytes per launch of the dominant kernel, stamped with the kernel sources they were measured on
python3 $REPO/tools/make_traffic.py $OUT $EXTRA > $OUT/traffic_entry.json
rm -rf $RAW
du -sh $OUT; ls $OUT
