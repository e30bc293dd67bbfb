#!/bin/bash
# MFMA utilisation of gemm_tn_stream_kernel (SQ counters), three shapes x 10 launches -> profiles/r02_c_tn_stream_pmc.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tnpmc; rm -rf $O; mkdir -p $O; cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $O/p1 -o p1 -- python $R/scripts/pmc_tn_stream_target.py > $O/p1.log 2>&1
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/tnpmc"
lines = []
for f in glob.glob(O + "/p1/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "gemm_tn_stream_kernel" in r["Kernel_Name"]]
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        by[(r["Kernel_Name"].split("gemm_tn_stream_kernel")[1][:14], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(by.items()):
        m = {n: sum(v) / len(v) for n, v in c.items()}
        util = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024) if m.get("GRBM_GUI_ACTIVE") else float("nan")
        lines.append("gemm_tn_stream_kernel%s grid %s: launches %d  GRBM_GUI_ACTIVE %.4g  SQ_VALU_MFMA_BUSY_CYCLES %.4g  MfmaUtil %.1f %%  SQ_WAIT_ANY/SQ_WAVE_CYCLES %.2f  SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES %.2f  MFMA_MOPS_F32 %.4g"
                     % (k[0], k[1], len(c["GRBM_GUI_ACTIVE"]), m["GRBM_GUI_ACTIVE"], m["SQ_VALU_MFMA_BUSY_CYCLES"], 100 * util,
                        m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_INSTS_VALU_MFMA_MOPS_F32"]))
open(O + "/tn_stream_pmc.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $O -name "*.csv" -size +1M -delete
