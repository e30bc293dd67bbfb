"""VGPR / spill / LDS figures per kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).

    python scripts/kernel_resources.py limb_gemm [extra hipcc flags]
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tf_gnn_samples_amd._build import HIPCC_FLAGS  # noqa: E402


def main():
    stem = sys.argv[1]
    src = ROOT / "tf_gnn_samples_amd" / "csrc" / (stem + ".hip")
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(["hipcc", *HIPCC_FLAGS, *sys.argv[2:], "-c", str(src), "-o", d + "/o.o",
                            "-Rpass-analysis=kernel-resource-usage"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout)
        sys.exit(1)
    cur = None
    rows = []
    for line in r.stdout.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|"
                      r"VGPRs Spill|SGPRs Spill):\s*(\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], stdout=subprocess.PIPE, text=True).stdout.strip()}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" [")[0]] = v
    for c in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", c["name"])
        name = re.sub(r"\(.*", "", name)
        print("%-48s vgpr %4s agpr %3s scratch %4s spill %3s lds %7s occ %s" % (
            name[:48], c.get("VGPRs"), c.get("AGPRs"), c.get("ScratchSize"), c.get("VGPRs Spill"), c.get("LDS Size"), c.get("Occupancy")))


if __name__ == "__main__":
    main()
