"""Build libptamd.so (HIP, gfx950 only) in-tree with hipcc.

    python -m protein_transformer_amd.build [--force] [--verbose]

The shared object lands next to the sources in `protein_transformer_amd/csrc/` so
that it travels with the repository snapshot to the GPU box.
"""
import argparse
import glob
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
# PTAMD_BUILD_TAG=<tag>: an ablation build (with PTAMD_EXTRA_FLAGS) next to the product library, loaded by setting
# PTAMD_LIB_TAG=<tag> (protein_transformer_amd/_lib.py); the product library and its objects are left alone
_TAG = os.environ.get("PTAMD_BUILD_TAG", "")
LIB = os.path.join(CSRC, f"libptamd_{_TAG}.so" if _TAG else "libptamd.so")
OBJ_DIR = os.path.join(CSRC, f"build_{_TAG}" if _TAG else "build")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", f"-I{INCLUDE}"]
FLAGS += os.environ.get("PTAMD_EXTRA_FLAGS", "").split()      # ablation builds (e.g. -DPT_NO_MIX_SPLIT), part of the object digest


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libptamd.so cannot be built")
    return exe


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _digest(path, extra):
    h = hashlib.sha1()
    h.update(" ".join(FLAGS).encode())
    with open(path, "rb") as f:
        h.update(f.read())
    for e in extra:
        with open(e, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def gemm_source_digest():
    """SHA-1 over the sources of the GEMM kernels: what a stored PMC traffic record (profiles/rNN/*_gemm_hbm_traffic.json)
    names as the code it was measured on; bench.py reports a record only while the digest matches the tree."""
    h = hashlib.sha1()
    names = sorted(glob.glob(os.path.join(CSRC, "gemm*")) + [os.path.join(CSRC, n) for n in ("hp_format.h", "split_bf16.h", "common.h")])
    for n in names:
        if os.path.isfile(n):
            h.update(os.path.basename(n).encode())
            with open(n, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h")))
    cc = hipcc()
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        stamp = obj + ".sha1"
        dig = _digest(src, headers)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        lang = ["-x", "hip"] if src.endswith(".hip") else []
        jobs.append((src, obj, stamp, dig, [cc, *FLAGS, *lang, "-c", src, "-o", obj]))

    def run(job):
        src, obj, stamp, dig, cmd = job
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip() and verbose:
            print(r.stderr)
        with open(stamp, "w") as f:
            f.write(dig)
        return src

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(LIB):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    print(build(a.force, a.verbose))
    sys.exit(0)
