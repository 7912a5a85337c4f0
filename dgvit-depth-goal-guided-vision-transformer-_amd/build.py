"""Build libdgvit_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python dgvit-depth-goal-guided-vision-transformer-_amd/build.py [--force]

Objects are rebuilt only when their source (or a header) is newer; the .so travels to the GPU box with
the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libdgvit_hip.so")
SOURCES = ["gemm.hip", "norm.hip", "attention.hip", "embed.hip", "conv.hip", "optim.hip", "profile.hip", "preprocess.hip", "gemm_bf16.hip",
           "attention_bf16.hip", "misc_bf16.hip", "heads.hip", "frame.hip", "dgvit_api.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "kernels.h"), os.path.join(CSRC, "bf16.h"), os.path.join(CSRC, "small_mma.h"), os.path.join(INCLUDE, "dgvit_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-fvisibility=hidden", "-I", INCLUDE]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(CSRC, src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    subprocess.run(["hipcc", *FLAGS, "-c", path, "-o", obj], check=True)
    return obj


def build(force=False, verbose=True):
    hipcc = subprocess.run(["which", "hipcc"], capture_output=True, text=True).stdout.strip()
    if not hipcc:
        raise RuntimeError("hipcc not found on PATH; libdgvit_hip.so cannot be built")
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in SOURCES]
    todo = [s for s, o in zip(SOURCES, objs) if force or _stale(o, [os.path.join(CSRC, s), *HEADERS, __file__])]
    if todo:
        if verbose:
            print("hipcc gfx950:", " ".join(todo), flush=True)
        with ThreadPoolExecutor(max_workers=min(4, len(todo))) as ex:
            list(ex.map(_compile, todo))
    if todo or _stale(LIB, objs):
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], check=True)
        if verbose:
            print("linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
