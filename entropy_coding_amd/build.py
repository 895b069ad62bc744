"""hipcc build driver for libcabac_hip.so (gfx950 only, in-tree so that the .so travels to the GPU box).

Every source is compiled to its own object under build/ (git-ignored, gpurun-ignored), in parallel, and only when it
or a header changed; the objects are then linked into the in-tree library."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(PKG, "libcabac_hip.so")
OBJ = os.path.join(ROOT, "build", "obj")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-I" + INCLUDE, "-I" + CSRC, "-I" + HOST]
FLAGS += os.environ.get("CABAC_EXTRA_FLAGS", "").split()   # experiments only (e.g. -DCABAC_PARSE_PROFILE)


def library_path():
    return LIB


def _sources():
    srcs = []
    for d in (CSRC, HOST):
        if os.path.isdir(d):
            for f in sorted(os.listdir(d)):
                if f.endswith((".hip", ".cpp")):
                    srcs.append(os.path.join(d, f))
    return srcs


def _headers():
    hdr = []
    for d in (CSRC, HOST, INCLUDE):
        if os.path.isdir(d):
            hdr += [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".h", ".hpp"))]
    return hdr


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _sources() + _headers())


def build_library(force=False, verbose=False):
    """Compile every HIP/C++ source of the product into libcabac_hip.so for gfx950."""
    if not force and not is_stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if os.path.exists(LIB):
            return LIB  # e.g. a box without the toolchain: use the prebuilt in-tree library
        raise RuntimeError("hipcc not found and no prebuilt libcabac_hip.so")
    import hashlib
    obj_dir = OBJ + "-" + hashlib.md5(" ".join(FLAGS).encode()).hexdigest()[:8]   # objects of another flag set are not reused
    os.makedirs(obj_dir, exist_ok=True)
    newest_header = max([os.path.getmtime(p) for p in _headers()] + [os.path.getmtime(__file__)])
    jobs = []
    for src in _sources():
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(len(jobs), 1))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(obj_dir, os.path.basename(s) + ".o") for s in _sources()]
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB + ".tmp"])
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    import sys
    print(build_library(force="--force" in sys.argv, verbose=True))
