#!/bin/bash
# MFMA utilisation of the opt-in fused aggregate -> MFMA-transform kernel next to the two-kernel default it competes with
# (seg_reduce into the (target, type) buckets + library GEMM), one RGCN layer forward on the C2 batch.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fusedpmc; rm -rf $O; mkdir -p $O; cd /tmp
RELGNN_FUSED_MFMA=1 timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p1 -o p1 -- python $R/scripts/exp_agg_first.py > $O/p1.log 2>&1
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/fusedpmc"
lines = []
for f in glob.glob(O + "/p1/**/*counter_collection.csv", recursive=True):
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "agg_transform" in k: k = "agg_transform" + k.split("agg_transform")[1][:28]
        elif k.startswith("Cijk"): k = "library GEMM " + k[k.find("_MT"):k.find("_MT") + 16]
        elif "seg_reduce_wave" in k: k = "seg_reduce_wave_kernel"
        else: continue
        by[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(by.items()):
        m = {n: sum(v) / len(v) for n, v in c.items()}
        if not m.get("GRBM_GUI_ACTIVE"): continue
        util = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)
        lines.append("%-44s launches %4d  GRBM_GUI_ACTIVE/8 %.4g cycles  MFMA_BUSY %.4g  MfmaUtil %5.1f %%  WAIT_ANY/WAVE_CYCLES %.2f"
                     % (k, len(c["GRBM_GUI_ACTIVE"]), m["GRBM_GUI_ACTIVE"] / 8, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), 100 * util,
                        m.get("SQ_WAIT_ANY", 0.0) / max(m.get("SQ_WAVE_CYCLES", 1.0), 1.0)))
open(O + "/fused_pmc.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
tail -5 $O/p1.log
find $O -name "*.csv" -size +1M -delete
