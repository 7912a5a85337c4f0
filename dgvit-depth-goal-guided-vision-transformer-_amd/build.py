"""Build libdgvit_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python dgvit-depth-goal-guided-vision-transformer-_amd/build.py [--force] [--report]

Objects are rebuilt only when their source (or a header) is newer; the .so travels to the GPU box with
the repo snapshot (it is git-ignored, not gpurun-ignored).

Register audit: every translation unit is compiled with -Rpass-analysis=kernel-resource-usage; the per-kernel figures
(VGPRs, AGPRs, scratch bytes per lane, spills, LDS, occupancy) are kept beside the object as <name>.resources.json, and the
build FAILS when a kernel uses scratch memory or spills registers unless it is named in SPILL_ALLOW below (with the reason).
`--report` prints the table.
"""
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libdgvit_hip.so")
LIB_DIAG = os.path.join(HERE, "libdgvit_hip_diag.so")   # the same sources with -DDGVIT_DIAG: knobs, stamps, experiments (tools/, A/B tests)
SOURCES = ["gemm.hip", "norm.hip", "attention.hip", "embed.hip", "conv.hip", "optim.hip", "profile.hip", "preprocess.hip", "gemm_bf16.hip",
           "gemm_bf16_stream.hip", "attention_bf16.hip", "misc_bf16.hip", "heads.hip", "block.hip", "dgvit_api.hip"]
DIAG_ONLY_SOURCES = ["frame.hip"]    # experiments that are not part of the product library
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "kernels.h"), os.path.join(CSRC, "bf16.h"), os.path.join(CSRC, "small_mma.h"),
           os.path.join(CSRC, "knobs.h"), os.path.join(INCLUDE, "dgvit_hip.h"), os.path.join(INCLUDE, "dgvit_hip_diag.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-fvisibility=hidden", "-I", INCLUDE, "-Rpass-analysis=kernel-resource-usage"]

# Kernels that may use scratch memory (regular expressions on the demangled-free mangled name), each with its reason.
# Everything else must have ScratchSize == 0 and no spills, or the build stops.
SPILL_ALLOW = {
}

_REMARK = re.compile(r"remark: (?:\s*)(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|"
                     r"VGPRs Spill|LDS Size \[bytes/block\]): (\S+)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def parse_resources(stderr_text):
    """[{name, sgprs, vgprs, agprs, scratch, occupancy, sgpr_spill, vgpr_spill, lds}] from the resource-usage remarks."""
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
            "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
            "LDS Size [bytes/block]": "lds"}
    out, cur = [], None
    for m in _REMARK.finditer(stderr_text):
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            out.append(cur)
        elif cur is not None:
            cur[keys[k]] = int(v)
    return out


def _obj(src, diag):
    return os.path.join(CSRC, "diag" if diag else "", src.replace(".hip", ".o"))


def _compile(job):
    src, diag = job
    obj = _obj(src, diag)
    os.makedirs(os.path.dirname(obj), exist_ok=True)
    path = os.path.join(CSRC, src)
    r = subprocess.run(["hipcc", *FLAGS, *(["-DDGVIT_DIAG"] if diag else []), "-c", path, "-o", obj], capture_output=True, text=True)
    other = "\n".join(l for l in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l and not re.match(r"^\s+\d* *\|", l)
                      and "remarks generated" not in l and "remark generated" not in l)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        raise RuntimeError(f"hipcc failed on {src}")
    if other.strip():
        sys.stderr.write(other + "\n")
    res = parse_resources(r.stderr)
    with open(obj.replace(".o", ".resources.json"), "w") as f:
        json.dump(res, f, indent=0)
    return obj


def audit(sources=SOURCES, diag=False):
    """(all kernels, offenders): offenders use scratch / spill and are not allow-listed."""
    kernels, bad = [], []
    for s in sources:
        p = _obj(s, diag).replace(".o", ".resources.json")
        if not os.path.exists(p):
            continue
        for k in json.load(open(p)):
            k["file"] = s
            kernels.append(k)
            if k.get("scratch", 0) or k.get("vgpr_spill", 0):     # (SGPR "spills" go to VGPR lanes: no memory traffic, reported only)
                if not any(re.search(pat, k["name"]) for pat in SPILL_ALLOW):
                    bad.append(k)
    return kernels, bad


def _demangle(names):
    try:
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return r.stdout.splitlines()
    except Exception:
        return names


def build(force=False, verbose=True, check_spills=True, diag=True):
    """Compile and link libdgvit_hip.so (the product) and, with diag=True, libdgvit_hip_diag.so (same sources, -DDGVIT_DIAG).
    The register audit gates the PRODUCT library; the diagnostic library's experiments may spill (they are reported, not fatal)."""
    hipcc = subprocess.run(["which", "hipcc"], capture_output=True, text=True).stdout.strip()
    if not hipcc:
        raise RuntimeError("hipcc not found on PATH; libdgvit_hip.so cannot be built")
    jobs = [(s, False) for s in SOURCES] + ([(s, True) for s in SOURCES + DIAG_ONLY_SOURCES] if diag else [])
    todo = [(s, d) for s, d in jobs if force or _stale(_obj(s, d), [os.path.join(CSRC, s), *HEADERS, __file__])
            or not os.path.exists(_obj(s, d).replace(".o", ".resources.json"))]
    if todo:
        if verbose:
            print("hipcc gfx950:", " ".join(s + ("[diag]" if d else "") for s, d in todo), flush=True)
        todo.sort(key=lambda j: -os.path.getsize(os.path.join(CSRC, j[0])))     # longest compiles first
        with ThreadPoolExecutor(max_workers=min(7, len(todo))) as ex:
            list(ex.map(_compile, todo))
    if check_spills:
        kernels, bad = audit()
        if bad:
            names = _demangle([k["name"] for k in bad])
            msg = "\n".join(f"  {k['file']}: {n}: scratch {k.get('scratch', 0)} B/lane, VGPR spills {k.get('vgpr_spill', 0)}, "
                            f"SGPR spills {k.get('sgpr_spill', 0)} (VGPRs {k.get('vgprs')}, AGPRs {k.get('agprs')})" for k, n in zip(bad, names))
            raise RuntimeError("kernels with register spills / scratch memory (fix them or allow-list them in build.py with a reason):\n" + msg)
    for lib, d, srcs in ((LIB, False, SOURCES), (LIB_DIAG, True, SOURCES + DIAG_ONLY_SOURCES)):
        if d and not diag:
            continue
        objs = [_obj(s, d) for s in srcs]
        if _stale(lib, objs):
            subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], check=True)
            if verbose:
                print("linked", lib, flush=True)
    return LIB


def report(diag=False):
    kernels, bad = audit(SOURCES + (DIAG_ONLY_SOURCES if diag else []), diag)
    names = _demangle([k["name"] for k in kernels])
    for k, n in sorted(zip(kernels, names), key=lambda kn: (-kn[0].get("scratch", 0), kn[0]["file"], kn[1])):
        print(f"{k['file']:20s} vgpr {k.get('vgprs', 0):3d} agpr {k.get('agprs', 0):3d} sgpr {k.get('sgprs', 0):3d} scratch {k.get('scratch', 0):4d} "
              f"spill {k.get('vgpr_spill', 0):3d} lds {k.get('lds', 0):6d} occ {k.get('occupancy', 0)}  {n[:150]}")
    print(f"{len(kernels)} kernels, {len(bad)} with scratch / spills outside the allow list")


if __name__ == "__main__":
    if "--report" in sys.argv:
        build(force="--force" in sys.argv, check_spills=False)
        report(diag="--diag" in sys.argv)
    else:
        build(force="--force" in sys.argv)
