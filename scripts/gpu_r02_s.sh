#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02s; rm -rf $O; mkdir -p $O; cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -o p1 -- python $R/scripts/pmc_gemm_target.py > $O/p1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/p2 -o p2 -- python $R/scripts/pmc_gemm_target.py > $O/p2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r02s"
for p in ("p1", "p2"):
    for f in glob.glob(O + "/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "gemm_f32_kernel" in k: k = "own:" + k.split("gemm_f32_kernel")[1][:22]
            elif k.startswith("Cijk"): k = "lib:" + k[60:90]
            else: continue
            acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print("%-40s %-32s n=%3d mean=%.4g" % (k, c, len(v), sum(v) / len(v)))
PY
find $O -name "*.csv" -size +2M -delete
