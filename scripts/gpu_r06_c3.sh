#!/bin/bash
# round 6, C3: the per-edge-type weight gradients as one streaming product (relgnn_gemm_tn_stream_blocks_f32) — its tests, the product
# alone against the five it replaces, C3 before / after through bench_other.py, and the step's kernel sequence
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_c3; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_lib_gemm.py tests/test_gpu_limb_gemm.py tests/test_gpu_reference_run.py tests/test_gpu_layers.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
python - <<'PY' 2>&1 | tee $O/product.txt
import torch
from tf_gnn_samples_amd import dense as DN
dev = torch.device("cuda:0")
for V in (49986, 20000):
    a = torch.rand((V, 128), device=dev) * 2 - 1
    g = (torch.rand((V, 640), device=dev) * 2 - 1) * 0.05
    def timed(fn, iters=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    print(V, "five products %.1f us" % timed(lambda: [DN.matmul_tn_splitk(a, g[:, l * 128:(l + 1) * 128]) for l in range(5)]),
          "one product into blocks %.1f us" % timed(lambda: DN.tn_stream_blocks(a, g, 5)))
PY
for i in 1 2; do timeout 600 python bench_other.py C3 2>/dev/null | cut -c1-400 | tee -a $O/c3.jsonl; done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o k -- python $R/bench_other.py C3 > $O/lines.jsonl 2> $O/err.txt
f=$(find $O/t -name "*kernel_trace.csv" | head -1); python $R/scripts/step_sequence.py $f 400 > $O/c3_step_sequence.txt
f=$(find $O/t -name "*kernel_stats.csv" | head -1); cp $f $O/c3_kernel_stats.csv
rm -rf $O/t
head -50 $O/c3_step_sequence.txt
