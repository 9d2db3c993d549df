# B-operand loads of dsp_lstm_kernel with the nt cache hint (variants/libdsp_B.so) vs in-tree: per-launch time and L2 misses
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash tools/experiments/r4_lib_ab.sh 65536 2>&1 | grep -v amdgpu.ids | grep "identical\|==\|lstm_comb\|sum"
cd /tmp && export TMPDIR=/tmp
for lib in "" $GRAFT_REPO_ROOT/variants/libdsp_B.so; do
  for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    N=$(echo $C | tr ' ' '_'); rm -rf /tmp/nt_$N
    DSP_AMD_LIB=$lib rocprofv3 --pmc $C --output-format csv -d /tmp/nt_$N -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_alt > /dev/null 2>&1
    echo "== lib ${lib:-in-tree} $C"; python3 $GRAFT_REPO_ROOT/tools/summarize_prof.py pmc /tmp/nt_$N /tmp/nt_$N.txt; grep "dsp_lstm_kernel<0, 1, 0>" /tmp/nt_$N.txt
  done
done
