#!/bin/bash
# Profile recipe for the extraction stage (GPU box): kernel trace + stats, then separate PMC passes (never combined
# with tracing domains).  Summaries under gpurun_out/prof_extract; raw traces are deleted on the box.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_extract
RAW=/tmp/prof_raw_extract
READS=${1:-512}
rm -rf $OUT $RAW; mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
python3 $REPO/tools/bench_extract.py $READS 8000 > $OUT/bench_extract.jsonl 2> $OUT/bench_extract.err
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o ext -- python3 $REPO/tools/bench_extract.py $READS 8000 > $OUT/trace.log 2>&1
find $RAW/trace -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $RAW/pmc_$N -o pmc -- python3 $REPO/tools/bench_extract.py $READS 8000 > $OUT/pmc_$N.log 2>&1
  python3 $REPO/tools/summarize_prof.py pmc $RAW/pmc_$N $OUT/pmc_$N.txt
done
rm -rf $RAW $OUT/*.log
du -sh $OUT; ls $OUT
