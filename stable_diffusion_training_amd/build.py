"""Build libsdtrain_hip.so (gfx950 only) in-tree with hipcc.  No JIT cache: the .so sits next to the sources so
that it travels with the repo snapshot to the GPU box."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libsdtrain_hip.so")
SOURCES = ["runtime.hip", "elementwise.hip", "optimizer.hip", "norm.hip", "gemm.hip", "attention.hip"]
EXTRA_FLAGS = {"optimizer.hip": ["-ffp-contract=off"],
               # attention keeps score tiles and running outputs in arch VGPRs: with the default AGPR form hipcc parks both in the
               # same accumulator registers and moves 96 values per tile through v_accvgpr_read/write
               "attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
HEADERS = [os.path.join(CSRC, "sdt_common.h"), os.path.join(os.path.dirname(CSRC), "..", "include", "sdt.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force=False, verbose=False):
    hipcc = _hipcc()
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result", "-Wno-unused-value"]
    base += os.environ.get("SDT_HIPCC_EXTRA", "").split()  # developer builds (e.g. -DSDT_ATTN_DBG ablations)
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + HEADERS):
            jobs.append((base + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o], src))

    def run(job):
        cmd, name = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {name}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return name

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
