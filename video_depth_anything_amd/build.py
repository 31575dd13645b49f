"""Compile csrc/*.hip into the in-tree shared library libvda_hip.so for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the
resulting .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libvda_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result",
         "-mllvm", "-amdgpu-mfma-vgpr-form"]   # MFMA results in VGPRs: no v_accvgpr copies around the softmax / epilogues
# Per-file overrides (appended, so they win). stitch.hip restates numpy float32 array arithmetic (a*b + c with two
# roundings): -ffp-contract=fast fuses in the backend regardless of source pragmas, so that file is built without it.
# The GEMM epilogues are scalar fp32 code per output element: the SLP vectoriser pairs neighbouring adds into
# v_pk_add_f32 (several times the issue cost of two v_add_f32 on gfx950) with the pairs misaligned against the
# fp16 packing, which costs more shuffles than arithmetic - off for those files.
PER_FILE = {"stitch.hip": ["-ffp-contract=off"]}
# attention: the softmax row sums are 32 scalar fp32 adds per tile; SLP packs 22 of them into v_pk_add_f32, which costs several
# times two v_add_f32 next to MFMAs (MI355X_MICROARCH.md, "price of one filler") in a loop that is VALU-bound.
PER_PREFIX = {"gemm": ["-fno-slp-vectorize"], "attention.hip": ["-fno-slp-vectorize"], "mlp_fused.hip": ["-fno-slp-vectorize"],
              # the dynamic tile draw of gemm8p_kernel.h must stay a plain returning atomic (see the comment there)
              "gemm8p": ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "vda.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    sp = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(sp), _deps_mtime()):
        return obj
    extra = list(PER_FILE.get(src, []))
    for prefix, fl in PER_PREFIX.items():
        if src.startswith(prefix):
            extra += fl
    cmd = [HIPCC] + FLAGS + extra + ["-c", sp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
    return obj


def build(verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(_compile, _sources()))
    if (not os.path.exists(LIB)) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    if verbose:
        print("built", LIB)
    return LIB


def build_host_demo(verbose=False):
    """examples/host_demo.cpp: a C++ host of the handle API, linked against libvda_hip.so (no Python, no torch)."""
    src = os.path.join(HERE, "..", "examples", "host_demo.cpp")
    exe = os.path.join(HERE, "host_demo")
    lib = build()
    if os.path.exists(exe) and os.path.getmtime(exe) > max(os.path.getmtime(src), os.path.getmtime(lib), _deps_mtime()):
        return exe
    cmd = [HIPCC, "-O2", "-std=c++17", src, "-I", os.path.join(HERE, "..", "include"), "-L", HERE, "-lvda_hip",
           "-Wl,-rpath,$ORIGIN", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"host_demo build failed:\n{r.stderr}")
    if verbose:
        print("built", exe)
    return exe


if __name__ == "__main__":
    build(verbose=True)
    build_host_demo(verbose=True)
