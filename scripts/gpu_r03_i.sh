#!/bin/bash
# round 3, trip i: ablation of the limb GEMM's k-loop (library rebuilt on the box with -DRELGNN_LIMB_ABLATE into a scratch copy);
# before that the tests and the per-shape benchmark of the shipped build
mkdir -p gpurun_out/r03i
timeout 600 python -m pytest tests/test_gpu_limb_gemm.py -x -q 2>&1 | tail -3
timeout 600 python scripts/bench_limb_gemm.py 2>/dev/null | tee gpurun_out/r03i/limb_gemm.jsonl | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if 'limb_tn_us' in d: print(d['shape'], '| limb_tn', d['limb_tn_us'], 'f32 route', d['f32_route_us'], '| err limb', '%.2e' % d['err_limb_vs_f64'], 'f32', '%.2e' % d['err_f32_vs_f64'])
        else: print(d['shape'], '| xf32', d['limb_xf32_us'], 'limb', d['limb_us'], 'lib', d['lib_f32_us'], 'panel', d['panel_f32_us'], '| err limb', '%.2e' % d['err_limb_vs_f64'], 'f32', '%.2e' % d['err_f32_vs_f64'])
"
cd tf_gnn_samples_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -DRELGNN_LIMB_ABLATE -c csrc/limb_gemm.hip -o /tmp/limb_ab.o || exit 1
cp librelgnn.so /tmp/librelgnn.keep
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o librelgnn.so $(ls build/*.o | grep -v limb_gemm.o) /tmp/limb_ab.o -lhipblaslt || exit 1
cd ..
for ab in 0 1 2 3 7 0; do
  RELGNN_LIMB_ABLATE=$ab timeout 300 python scripts/bench_limb_ablate.py 2>/dev/null
done | tee gpurun_out/r03i/limb_ablate.txt
cp /tmp/librelgnn.keep tf_gnn_samples_amd/librelgnn.so
