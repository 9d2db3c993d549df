# hand-off variants: in-tree (E 3/2/1 + early poll) vs B (E 1/1/1), C (E 3/2/1, no early poll), D (E 0: per-wave arrival at once), and round 4's protocol
for r in 1 2; do
for lib in "" variants/libdsp_B.so variants/libdsp_C.so variants/libdsp_D.so; do
  echo "== lib ${lib:-in-tree}"
  DSP_R5_MODES="auto:" DSP_AMD_LIB=${lib:+$PWD/$lib} python tools/experiments/r5_small_ab.py 512,1024,2048 300 2>&1 | grep -v amdgpu.ids | grep "^512\|^1024\|^2048\|identical"
done
echo "== in-tree, round 4's hand-off"
DSP_R5_MODES="old:DSP_LSTM_HANDOFF=0" python tools/experiments/r5_small_ab.py 512,1024,2048 300 2>&1 | grep -v amdgpu.ids | grep "^512\|^1024\|^2048\|identical"
done
