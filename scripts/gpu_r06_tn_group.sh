cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lib_gemm.py tests/test_gpu_gru_cell.py tests/test_gpu_reference_run.py tests/test_gpu_layers.py -m gpu -x -q 2>&1 | tail -4
python - <<'PY'
import torch
from tf_gnn_samples_amd import dense as DN
dev = torch.device("cuda:0")
V, u = 49986, 128
x, h, rh = (torch.rand((V, u), device=dev) * 2 - 1 for _ in range(3))
gxk = (torch.rand((V, 3 * u), device=dev) * 2 - 1) * 0.05
gq = gxk[:, 2 * u:].contiguous()
def timed(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
def separate():
    gK = DN.matmul_tn_splitk(x, gxk)
    gU = torch.empty((u, 3 * u), device=dev)
    DN.tn_stream_into(h, gxk[:, :2 * u], gU[:, :2 * u]); DN.tn_stream_into(rh, gq, gU[:, 2 * u:])
    return gK, gU, DN.column_sum(gxk)
def grouped():
    gK = torch.empty((u, 3 * u), device=dev); gU = torch.empty((u, 3 * u), device=dev); gb = torch.empty(3 * u, device=dev)
    DN.tn_stream_group([(x, gxk, gK), (h, gxk[:, :2 * u], gU[:, :2 * u]), (rh, gq, gU[:, 2 * u:])], colsum=gb)
    return gK, gU, gb
print("GRU weight gradients at V = 49986: separate %.1f us, one group %.1f us" % (timed(separate), timed(grouped)))
PY
for i in 1 2; do timeout 600 python bench_other.py C3 2>/dev/null | cut -c1-260; done
