"""Build librelgnn.so (the C-ABI HIP library, include/relgnn.h) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the
resulting tf_gnn_samples_amd/librelgnn.so travels with the repo snapshot to the GPU box.
"""
import os
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = PKG_DIR / "librelgnn.so"
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # product and add of `scale * row` then `acc + msg` are rounded separately, like the
    # reference's op chain (mul op, then segment-sum op); explicit fmaf() is used where wanted.
    "-ffp-contract=off",
    "-Wno-unused-value",
]


def _sources():
    return sorted(CSRC.glob("*.hip"))


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build librelgnn.so")
    return exe


def is_stale() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    deps = _sources() + list(CSRC.glob("*.h")) + [PKG_DIR.parent / "include" / "relgnn.h"]
    return any(d.stat().st_mtime > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> Path:
    """Compile every csrc/*.hip into one shared object.  Returns its path."""
    objdir = PKG_DIR / "build"
    objdir.mkdir(exist_ok=True)
    stems = {src.stem for src in _sources()}
    for stale in objdir.glob("*.o"):                  # objects of kernels that no longer have a source (removed experiments)
        if stale.stem not in stems:
            stale.unlink()
    if not force and not is_stale():
        return LIB_PATH
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in _sources():
        obj = objdir / (src.stem + ".o")
        objs.append(obj)
        newest_dep = max([src.stat().st_mtime] + [h.stat().st_mtime for h in CSRC.glob("*.h")]
                         + [(PKG_DIR.parent / "include" / "relgnn.h").stat().st_mtime])
        if not force and obj.exists() and obj.stat().st_mtime > newest_dep:
            continue
        cmd = [hipcc, *HIPCC_FLAGS, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, out.decode(errors="replace")))
    tmp = LIB_PATH.with_suffix(".so.tmp")
    # hipBLASLt: the plain library GEMMs of csrc/blaslt_gemm.hip (in a process that has imported torch the already loaded
    # copy with the same SONAME is the one that gets used)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(tmp), *map(str, objs), "-lhipblaslt"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stdout.decode(errors="replace"))
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    import sys
    print(build_library(force="--force" in sys.argv, verbose=True))
