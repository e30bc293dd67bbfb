#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the FiLM backward on the C2 batch, both routes of the
# by-source gradient: RELGNN_EDGE_BWD=emit (per-message gradients written, then gather-reduced) vs regather (default for
# the wave kernels).  Run through gpurun; output gpurun_out/edgebwd/edge_bwd_pmc.txt
export TMPDIR=/tmp RELGNN_CAPTURE=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/edgebwd; rm -rf $O; mkdir -p $O; cd /tmp
for MODE in emit regather; do
  for C in FETCH_SIZE WRITE_SIZE; do
    RELGNN_EDGE_BWD=$MODE timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${MODE}_$C -o k -- \
        python $R/scripts/bench_configs.py FILM > $O/${MODE}_$C.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, os, re, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/edgebwd"
lines = ["FiLM on the C2 batch (1 854 895 messages, D = 256): bytes per launch = 2*FETCH_SIZE*1024 (reads) / WRITE_SIZE*1024 (writes)"]
for mode in ("emit", "regather"):
    by = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(O + "/%s_%s/**/*counter_collection.csv" % (mode, c), recursive=True):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                m = re.search(r"(edge_\w+<[^>]*>|seg_reduce_wave_kernel<1, false, false[^>]*>)", k)
                if m and r["Counter_Name"] == c:
                    acc[m.group(1)].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                by[k][c] = (len(v), sum(v) / len(v))
    for k, d in sorted(by.items()):
        rd = 2 * d.get("FETCH_SIZE", (0, 0))[1] * 1024 / 1e6
        wr = d.get("WRITE_SIZE", (0, 0))[1] * 1024 / 1e6
        lines.append("%-9s %-58s launches %3d  read %8.1f MB  written %8.1f MB" % (mode, k, d.get("FETCH_SIZE", (0, 0))[0], rd, wr))
open(O + "/edge_bwd_pmc.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
find $O -name "*.csv" -size +1M -delete
