#!/bin/bash
# Build ablation variants of the kernels (compile-time -DABL_* switches) and time them with bench.py.
# Usage (on the build box):  bash tools/ablate.sh build ; then on the GPU box: bash tools/ablate.sh run
set -u
REPO=$(cd $(dirname $0)/.. && pwd)
C=$REPO/deepsignal_plant_amd/csrc
VARIANTS=${VARIANTS:-"BASE 3_NOA 3_NOB 3_NOBX 3_NOBH 3_NOCELL 3_NOA_NOB"}
if [ "$1" = build ]; then
  mkdir -p $REPO/gpurun_abl
  for V in $VARIANTS; do
    D=""; [ $V != BASE ] && D=$(echo "$V" | sed 's/^3//; s/_\([A-Z0-9]*\)/ -DABL3_\1/g')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $D -I$REPO/include -I$C -c $C/dsp_kernels.hip -o /tmp/abl_$V.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/gpurun_abl/libdsp_$V.so /tmp/abl_$V.o $C/_obj/dsp_capi.o $C/_obj/dsp_text.o -pthread &
  done
  wait
  ls -la $REPO/gpurun_abl
else
  for V in $VARIANTS; do
    echo "== $V"
    DSP_AMD_LIB=$REPO/gpurun_abl/libdsp_$V.so python3 $REPO/bench.py --steps 4 --warmup 1 --no_cpu_baseline 2>&1 | grep -o '"value": [0-9.]*\|"achieved": [0-9.]*\|"avg_launch_ms": [0-9.]*' | tr '\n' ' '
    echo
  done
fi
