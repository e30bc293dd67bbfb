#!/bin/bash
# HBM traffic (PMC FETCH_SIZE / WRITE_SIZE, separate passes) of the non-headline gather kernels: FiLM edge kernels and
# the D=128 group reduce on the C5 batch, RGAT kernels on the C2 batch.  Run through gpurun.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_edge_pmc
rm -rf $O; mkdir -p $O
cd /tmp
for CFG in C5 C4; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${CFG}_$C -o k -- \
        python $R/scripts/bench_configs.py $CFG > $O/${CFG}_$C.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof_edge_pmc"
want = ("edge_fwd_kernel", "edge_bwd_rows_kernel", "edge_fwd_wave", "edge_bwd_rows_wave", "seg_reduce_group_kernel",
        "seg_reduce_wave_kernel", "headw_reduce_kernel", "rgat_dz_kernel", "rgat_alpha_kernel", "rgat_scores")
lines = ["config,kernel,counter,launches,mean_KiB_per_launch"]
for cfg in ("C5", "C4"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(O + "/%s_%s/**/*counter_collection.csv" % (cfg, c), recursive=True):
            by = {}
            for r in csv.DictReader(open(f)):
                k = r.get("Kernel_Name", "")
                name = next((w for w in want if w in k), None)
                if name and r["Counter_Name"] == c:
                    # keep template arguments short
                    # ("void (anonymous namespace)::edge_fwd_kernel<32, 1, 0, false>(...)": the text before the first "(" is "void ")
                    short = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
                    by.setdefault(short, []).append(float(r["Counter_Value"]))
            for k, v in sorted(by.items()):
                lines.append("%s,%s,%s,%d,%.1f" % (cfg, k, c, len(v), sum(v) / len(v)))
open(O + "/edge_kernels_pmc.csv", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
for f in glob.glob(O + "/**/*kernel_trace.csv", recursive=True):
    os.remove(f)
PY
