#!/bin/bash
# Matrix-pipe / VALU / LDS counters of rgcn_fused_kernel next to the two kernels it replaces, on the C2 batch of
# scripts/bench_rgcn_fused.py (separate --pmc passes; MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_rgcn_fused; rm -rf $O; mkdir -p $O
cd /tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  d=$O/$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -o k -- python $R/scripts/bench_rgcn_fused.py > /dev/null 2>> $O/pmc.err
done
python - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_rgcn_fused"
agg = {}
for f in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "rgcn_fused" in k or "seg_reduce_wave" in k or "limb_gemm_kernel" in k:
            agg.setdefault((k.replace("void (anonymous namespace)::", "")[:44], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
lines = []
for (k, c), v in sorted(agg.items()):
    lines.append("%-46s %-26s n=%4d mean %.5g" % (k, c, len(v), sum(v) / len(v)))
by = {}
for (k, c), v in agg.items():
    by.setdefault(k, {})[c] = sum(v) / len(v)
for k, d in sorted(by.items()):
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("GRBM_GUI_ACTIVE"):
        lines.append("%-46s MfmaUtil = %.3f" % (k, d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)))
open(O + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
tail -3 $O/pmc.err
