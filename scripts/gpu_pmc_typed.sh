#!/bin/bash
# counters of the typed K = 128 product kernels (what holds them at ~240 us for 754 MB of traffic and 145 GFLOP of bf16 products?)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_typed; rm -rf $O; mkdir -p $O
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_STALL_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES"; do
  d=$O/$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -o k -- python $R/scripts/pmc_typed_target.py > /dev/null 2>> $O/pmc.err
done
python - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_typed"
agg = {}
for f in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "limb_gemm" in k or "panel_gemm" in k:
            agg.setdefault((k.replace("void (anonymous namespace)::", "")[:34], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
with open(O + "/summary.txt", "w") as f:
    for (k, c), v in sorted(agg.items()):
        f.write("%-36s %-30s n=%3d mean %.5g\n" % (k, c, len(v), sum(v) / len(v)))
print(open(O + "/summary.txt").read())
PY
tail -3 $O/pmc.err
