#!/bin/bash
# MFMA utilisation evidence for the fused aggregate -> f32-MFMA transform kernel (csrc/rgcn_fused.hip, experimental,
# RELGNN_FUSED_MFMA=1): SQ counters of the kernel alone on the C2 batch.  Run through gpurun; PMC passes only
# (--kernel-trace), never with sys/hip tracing.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_fused
rm -rf $O; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $O/pmc -o k -- python $R/scripts/bench_fused.py > $O/pmc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o k -- python $R/scripts/bench_fused.py > $O/trace.log 2>&1
cd $R
tail -3 $O/trace.log
python - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof_fused"
out = []
for f in glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True):
    by = {}
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "rgcn_fused" in k or "Cijk" in k or "seg_reduce_wave" in k:
            name = "rgcn_fused_kernel" if "rgcn_fused" in k else ("hipblaslt_gemm" if "Cijk" in k else "seg_reduce_wave_kernel")
            by.setdefault((name, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(by.items()):
        out.append("%s,%s,%d,%.1f" % (k, c, len(v), sum(v) / len(v)))
print("kernel,counter,launches,mean_per_launch")
print("\n".join(out))
open(O + "/fused_mfma_pmc.csv", "w").write("kernel,counter,launches,mean_per_launch\n" + "\n".join(out) + "\n")
for f in glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True):
    os.system("head -6 %s | cut -c1-200" % f)
PY
