#!/bin/bash
# LDS-side counters of the limb GEMM kernels (is the LDS array the busy unit?)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_limb_lds; rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_WAIT_INST_[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*" | sort -u | tr '\n' ' ' > $O/avail.txt
for grp in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  d=$O/$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -o k -- python $R/scripts/pmc_limb_target.py > /dev/null 2>> $O/pmc.err
done
python - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_limb_lds"
agg = {}
for f in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "limb_gemm" in k or k.startswith("Cijk"):
            agg.setdefault((k.replace("void (anonymous namespace)::", "")[:40], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
with open(O + "/summary.txt", "w") as f:
    for (k, c), v in sorted(agg.items()):
        f.write("%-42s %-26s n=%3d mean %.4g\n" % (k, c, len(v), sum(v) / len(v)))
print(open(O + "/summary.txt").read())
PY
cat $O/avail.txt | cut -c1-1500; tail -3 $O/pmc.err
