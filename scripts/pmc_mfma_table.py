#!/usr/bin/env python
"""Matrix-pipe counters per kernel of any command: one rocprofv3 --kernel-trace --stats pass (average durations) and one --pmc pass
(SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_MFMA, GRBM_GUI_ACTIVE), joined by kernel name through bench.mfma_pass — the table the bench line's
roofline.mfma comes from, for other workloads (the C5 / C3 steps).   usage: pmc_mfma_table.py <command ...>   (run from /tmp)"""
import csv, glob, json, os, shutil, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench

exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
child = sys.argv[1:]
env = dict(os.environ, TMPDIR="/tmp")
tmp = tempfile.mkdtemp(prefix="relgnn_mfma_", dir="/tmp")
d = os.path.join(tmp, "trace")
r = subprocess.run([exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "k", "--", *child], cwd="/tmp", env=env,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, text=True)
files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
if r.returncode != 0 or not files:
    raise SystemExit("kernel trace failed (rc %d): %s" % (r.returncode, r.stdout[-400:]))
rows = list(csv.DictReader(open(files[0])))
m = bench.mfma_pass(exe, child, env, tmp, rows, 900)
shutil.rmtree(tmp, ignore_errors=True)
if "error" in m:
    raise SystemExit(m["error"])
print("# %s" % " ".join(child))
print("# busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024); PFLOP/s = SQ_INSTS_MFMA x 32768 / average duration (16-bit 32x32x16 MFMAs)")
print("%-72s %6s %9s %9s %8s %9s %7s" % ("kernel", "calls", "avg_us", "busy_frac", "PFLOP/s", "of 2.5 PF", "share"))
for k in m["kernels"]:
    print("%-72s %6d %9.1f %9.3f %8.3f %9.3f %7.3f" % (k["kernel"][:72], k["calls"], k["avg_kernel_us"], k["busy_frac"], k["bf16_PFLOPs"],
                                                       k["frac_of_bf16_peak"], k["share_of_mfma_kernel_time"]))
