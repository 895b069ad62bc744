"""hipcc build driver for libcabac_hip.so (gfx950 only, in-tree so that the .so travels to the GPU box)."""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(PKG, "libcabac_hip.so")


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


def _deps():
    deps = list(_sources())
    for d in (CSRC, HOST, INCLUDE):
        if os.path.isdir(d):
            deps += [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".h", ".hpp"))]
    return deps


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _deps())


def build_library(force=False, verbose=False):
    """Compile every HIP/C++ source of the product into libcabac_hip.so for gfx950."""
    if not force and not is_stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if os.path.exists(LIB):
            return LIB  # e.g. a box without the toolchain: use the prebuilt in-tree library
        raise RuntimeError("hipcc not found and no prebuilt libcabac_hip.so")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
           "-I" + INCLUDE, "-I" + CSRC, "-I" + HOST] + _sources() + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
