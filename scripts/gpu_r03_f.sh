#!/bin/bash
# round 3, GPU session F: (1) the N > 1 path of bench.py as 2 ranks sharing the one GPU over gloo (launch-path check of the new
# per-rank statistics, C2 and C5); (2) fabric-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the FiLM edge kernels and
# the D = 128 group reduce on the C5 batch, RGAT kernels on the C2 batch; (3) the fixed gradient-parity test
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; rm -rf $O; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_baseline_size.py -q -s -k "gradients or report" > $O/t_grad.txt 2>&1; echo "rc=$?" >> $O/t_grad.txt
RELGNN_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-extras > $O/bench_2ranks_one_gpu_gloo.json 2> $O/bench2.err; echo "rc=$?" >> $O/bench2.err
RELGNN_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --config C5 --steps 6 --warmup 2 --no-roofline --no-cpu-baseline --no-extras > $O/bench_c5_2ranks_one_gpu_gloo.json 2>> $O/bench2.err; echo "rc=$?" >> $O/bench2.err
cd /tmp
for CFG in C5 C4; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${CFG}_$C -o k -- python $R/bench_other.py $CFG > $O/${CFG}_$C.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r03f"
want = ("edge_fwd_kernel", "edge_bwd_rows_kernel", "seg_reduce_group_kernel", "headw_reduce_kernel", "rgat_dz_kernel", "rgat_alpha_kernel",
        "rgat_scores", "panel_gemm_kernel")
lines = ["config,kernel,counter,launches,mean_KiB_per_launch"]
for cfg in ("C5", "C4"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(O + "/%s_%s/**/*counter_collection.csv" % (cfg, c), recursive=True):
            by = {}
            for r in csv.DictReader(open(f)):
                k = r.get("Kernel_Name", "")
                name = next((w for w in want if w in k), None)
                if name and r["Counter_Name"] == c:
                    short = k.split("(")[0].split("::")[-1][:60]
                    by.setdefault(short, []).append(float(r["Counter_Value"]))
            for k, v in sorted(by.items()):
                lines.append("%s,%s,%s,%d,%.1f" % (cfg, k, c, len(v), sum(v) / len(v)))
open(O + "/edge_kernels_pmc.csv", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
for f in glob.glob(O + "/**/*kernel_trace.csv", recursive=True):
    os.remove(f)
PY
tail -4 $O/t_grad.txt; cut -c1-200 $O/bench_2ranks_one_gpu_gloo.json; python -c "
import json
for f in ('bench_2ranks_one_gpu_gloo.json','bench_c5_2ranks_one_gpu_gloo.json'):
    d=json.load(open('$O/'+f)); print(f, d['ms_per_step'], d['world_size'], d['backend'], d['per_rank'], d['step_edge_imbalance_max_over_mean'], d['gradient_allreduce_bytes'])
"; tail -3 $O/bench2.err
