#!/bin/bash
# round 4: the pipeline with the rows parsed on the GPU vs on the host, by host threads; the host feed of N ranks
export DSP_WORK=/tmp/dsp_pipe DSP_BENCH_NO_DSPF=1
out=gpurun_out/r4
mkdir -p $out
python tools/make_tsv.py /tmp/dsp_pipe/feat_4000000.tsv 4000000 > /dev/null 2>&1 || mkdir -p /tmp/dsp_pipe
: > $out/pipeline_cli_parse.jsonl
for mode in host device; do
  DSP_PARSE_ON=$mode DSP_BENCH_THREADS=1,2,4,10 python tools/bench_pipeline.py 4000000 2>/dev/null | grep '^{' >> $out/pipeline_cli_parse.jsonl
done
: > $out/feed_ranks.jsonl
python tools/make_tsv.py /tmp/dsp_pipe/feat_3200000.tsv 3200000 > /dev/null 2>&1
for mode in host device; do
  for r in 1 2 4 8; do
    # (400,000 rows per rank: a dozen blocks each, so that staging / formatting / the counting pass overlap as they do in a run)
    python tools/bench_feed.py --ranks $r --rows 3200000 --parse_on $mode 2>/dev/null | grep '^{' >> $out/feed_ranks.jsonl
  done
done
python - <<'PY'
import json
for f in ("gpurun_out/r4/pipeline_cli_parse.jsonl", "gpurun_out/r4/feed_ranks.jsonl"):
    print(f)
    for l in open(f):
        d = json.loads(l)
        if "parse_threads" in d:
            print("  parse_on %-6s -p %2d: %.2f s  %.3f M sites/s" % (d["parse_on"], d["parse_threads"], d["call_mods_s"], d["sites_per_s"] / 1e6))
        else:
            print("  %-5s ranks %d x %d threads: %.2f M rows/s all ranks = %.2f GPUs fed; cpu %.2f us/row -> %.2f host threads per rank at the GPU's rate" % (
                "dev" if "GPU (" in d["what"] or "parsed on the GPU" in d["what"] else "host", d["ranks"], d["threads_per_rank"], d["rows_per_s_all_ranks"] / 1e6,
                d["ranks_fed_at_full_gpu_rate"], d["cpu_us_per_row"], d["host_threads_per_rank_at_full_gpu_rate"]))
PY
