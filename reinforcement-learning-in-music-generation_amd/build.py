"""Build libcwlt.so (every HIP kernel + the C-ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU.  Output: <package>/libcwlt.so (git-ignored, travels to the GPU
box with the gpurun snapshot).  Objects are cached under csrc/.obj keyed on source mtime.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, ".obj")
LIB = os.path.join(PKG, "libcwlt.so")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")          # the public C-ABI header (cwlt.h)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: MFMA accumulators stay in VGPRs.  The kernels that feed accumulators back as operands
# (scan states, score tiles) otherwise pay a v_accvgpr_read/write per element per use (~20 % of the scan's VALU
# instructions) and use MORE registers in total (146 VGPR + 96 AGPR vs 180 VGPR for the forward scan).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-mfma-vgpr-form", "-I", INCLUDE]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(INCLUDE, "cwlt.h"))
    headers.append(os.path.abspath(__file__))      # a change of flags rebuilds everything
    jobs = []
    objs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        # -no-hip-rt: do NOT record a DT_NEEDED on /opt/rocm's libamdhip64.so.7.  PyTorch ships its own HIP
        # runtime (torch/lib/libamdhip64.so); two runtimes in one process do not share streams or
        # allocations.  The hip* symbols are bound at load time to the runtime already in the process
        # (_lib.load() puts torch's in the global scope first; a C host links libamdhip64 itself).
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-no-hip-rt", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
