#!/bin/bash
# librelgnn with one kernel file rebuilt under a diagnostic macro (s_memtime stamps): scripts/build_timing_variant.sh rgcn_fused RELGNN_FUSED_TIMING
# -> tf_gnn_samples_amd/build/librelgnn_<stem>_timing.so (git-ignored; travels to the GPU box)
set -e
cd "$(dirname "$0")/../tf_gnn_samples_amd"
stem=$1; macro=$2
python -c "from tf_gnn_samples_amd._build import build_library; build_library()" 2>/dev/null || (cd .. && python -c "from tf_gnn_samples_amd._build import build_library; build_library()")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -D$macro -c csrc/$stem.hip -o build/${stem}_timing.o
objs=$(ls build/*.o | grep -v "_timing.o" | grep -v "build/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/librelgnn_${stem}_timing.so $objs build/${stem}_timing.o -lhipblaslt
echo built tf_gnn_samples_amd/build/librelgnn_${stem}_timing.so
