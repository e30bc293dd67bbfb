#!/bin/bash
# SQ counters of the wave edge kernels on the C2 batch: GNN-Edge-MLP0 (GELU per message) next to GNN-FiLM (ReLU).
# Shows how much of the wave-cycles is VALU issue (the GELU kernels were ALU-bound with the library erff) and how much
# is parked on memory.  Run through gpurun; output gpurun_out/edgeact/edge_act_pmc.txt
export TMPDIR=/tmp RELGNN_CAPTURE=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/edgeact; rm -rf $O; mkdir -p $O; cd /tmp
for CFG in MLP0 FILM; do
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d $O/$CFG -o p -- python $R/scripts/bench_configs.py $CFG > $O/$CFG.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, os, collections, re
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/edgeact"
lines = []
for cfg in ("MLP0", "FILM"):
    for f in glob.glob(O + "/%s/**/*counter_collection.csv" % cfg, recursive=True):
        by = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "edge_" not in k: continue
            m_ = re.search(r"(edge_\w+<[^>]*>)", k)
            k = m_.group(1) if m_ else k[:44]
            by[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in sorted(by.items()):
            m = {n: sum(v) / len(v) for n, v in c.items()}
            wc = max(m.get("SQ_WAVE_CYCLES", 1.0), 1.0)
            lines.append("%-5s %-44s launches %3d  GUI_ACTIVE/8 %8.0f cyc  VALU insts %.3g  ACTIVE_VALU/WAVE_CYCLES %.3f  ACTIVE_ANY/WAVE_CYCLES %.3f  WAIT_ANY/WAVE_CYCLES %.3f  WAIT_INST_ANY/WAVE_CYCLES %.3f"
                         % (cfg, k, len(c["SQ_WAVE_CYCLES"]), m.get("GRBM_GUI_ACTIVE", 0) / 8, m.get("SQ_INSTS_VALU", 0), m.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                            m.get("SQ_ACTIVE_INST_ANY", 0) / wc, m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc))
open(O + "/edge_act_pmc.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $O -name "*.csv" -size +1M -delete
