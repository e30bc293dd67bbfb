#!/bin/bash
# round 3, trip i: ablation of the limb GEMM's k-loop (library rebuilt on the box with -DRELGNN_LIMB_ABLATE into a scratch copy)
mkdir -p gpurun_out/r03i

cd tf_gnn_samples_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -DRELGNN_LIMB_ABLATE -c csrc/limb_gemm.hip -o /tmp/limb_ab.o || exit 1
cp librelgnn.so /tmp/librelgnn.keep
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o librelgnn.so $(ls build/*.o | grep -v limb_gemm.o) /tmp/limb_ab.o -lhipblaslt || exit 1
cd ..
for ab in 0 100 101 102 1 2 3 7 0; do
  RELGNN_LIMB_ABLATE=$ab timeout 300 python scripts/bench_limb_ablate.py 2>/dev/null
done | tee gpurun_out/r03i/limb_ablate.txt
cp /tmp/librelgnn.keep tf_gnn_samples_amd/librelgnn.so
