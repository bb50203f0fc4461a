"""Build libctn_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so must travel with the tree."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libctn_hip.so")
SOURCES = ["ctn_api.hip", "ctn_gemm.hip", "ctn_tcn.hip", "ctn_bn.hip", "ctn_codec.hip", "ctn_loss.hip", "ctn_optim.hip",
           "ctn_block.hip"]
# -amdgpu-mfma-vgpr-form: MFMA results stay in VGPRs (unified file on gfx950), so the epilogues read them without
# v_accvgpr_read copies -- every VALU instruction serialises with the fp32 MFMAs (profiles/r02_a_mfma_probe.txt)
# -fno-slp-vectorize: keeps the compiler from packing adjacent scalar f32 adds / subs into v_pk_add_f32, which costs more
# beside MFMAs than the two scalar instructions (MI355X_MICROARCH.md, issue-cost table); b3 GEMMs 1-3 % faster, step -0.4 %
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"] + os.environ.get("CTN_EXTRA_HIPCC_FLAGS", "").split()


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile every .hip source for gfx950 and link libctn_hip.so next to this file."""
    hdrs = [os.path.join(CSRC, h) for h in ("ctn_common.h", "ctn_gemm_common.h", "ctn_gemm_b3.h", "ctn_gemm_ws.h")]
    objs, jobs = [], []
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, "build", os.path.basename(s).replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([_hipcc()] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
