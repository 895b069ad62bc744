"""Host AddressSanitizer / UBSan job for the C++ shim and the reference adapters (GPU sanitizers are not
available on the pool).  The host code — entropy_coding_amd/host, integration/reference_adapter.hpp and their test drivers — is
built once more under -fsanitize=address,undefined together with tests/csrc/cabac_hip_stub.cpp, a stand-in for the C ABI that
answers from the oracle (test infrastructure: the product has no CPU path), and the ordinary shim / adapter tests — the
GPU-marked ones included — are run against that build in a child interpreter with the sanitizer runtime preloaded.  What this
checks is the host code's memory behaviour (heap overruns, use after free, double frees, misaligned or overflowing
arithmetic); parity is the business of the -m gpu tests."""
import os
import subprocess
import sys

import pytest

import helpers as H

CSRC = os.path.join(H.ROOT, "tests", "csrc")
SAN_DIR = os.path.join(CSRC, "_san")
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
INC = ["-I" + os.path.join(H.ROOT, "include"), "-I" + os.path.join(H.ROOT, "entropy_coding_amd", "host"),
       "-I" + os.path.join(H.ROOT, "oracle"), "-I" + os.path.join(H.ROOT, "integration")]


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return os.path.realpath(p) if p and os.path.sep in p else None


def _stale(target, sources):
    return not os.path.exists(target) or os.path.getmtime(target) < max(os.path.getmtime(s) for s in sources)


def _build_shim_driver():
    os.makedirs(SAN_DIR, exist_ok=True)
    so = os.path.join(SAN_DIR, "libhost_shim_driver_san.so")
    cxx = [os.path.join(CSRC, "host_shim_driver.cpp"), os.path.join(CSRC, "cabac_hip_stub.cpp"),
           os.path.join(H.ROOT, "entropy_coding_amd", "host", "cabac_hip_host.cpp")]
    c = os.path.join(H.ROOT, "oracle", "cabac_oracle.c")
    hdr = [os.path.join(H.ROOT, "entropy_coding_amd", "host", "cabac_hip_host.hpp"), os.path.join(H.ROOT, "include", "cabac_hip.h")]
    if _stale(so, cxx + [c] + hdr):
        obj = os.path.join(SAN_DIR, "cabac_oracle_san.o")
        subprocess.check_call(["gcc", "-std=c11", "-D_POSIX_C_SOURCE=200809L", "-fPIC", "-pthread"] + SAN_FLAGS + INC + ["-c", c, "-o", obj])
        subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-shared", "-pthread"] + SAN_FLAGS + INC + cxx + [obj, "-o", so])
    return so


def _run_child(env_extra, test_file, timeout=900):
    asan = _libasan()
    if not asan:
        pytest.skip("no libasan for this gcc")
    env = dict(os.environ)
    env.update(env_extra)
    # libstdc++ beside it: the interpreter itself is not linked against it, and the runtime's __cxa_throw interceptor looks
    # its target up when it starts
    stdcxx = subprocess.run(["gcc", "-print-file-name=libstdc++.so"], capture_output=True, text=True).stdout.strip()
    env["LD_PRELOAD"] = asan + " " + os.path.realpath(stdcxx)
    # leaks are not the subject (the interpreter itself never frees everything); everything else aborts the child
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1:halt_on_error=1:allocator_may_return_null=1"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(H.ROOT, "tests", test_file), "-x", "-q", "-p", "no:cacheprovider",
                        "-m", "gpu or not gpu"], env=env, capture_output=True, text=True, timeout=timeout, cwd=H.ROOT)
    log = r.stdout[-6000:] + r.stderr[-6000:]
    assert "AddressSanitizer" not in log and "runtime error:" not in log, log
    assert r.returncode == 0, log
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], log   # nothing silently left out
    return r.stdout


def test_host_shim_under_asan_ubsan():
    """tests/test_host_shim.py — recorder, containers, Deferred / Immediate encode, pinned mirrors, decode replay, estimator,
    residual round trip — against the sanitized build."""
    so = _build_shim_driver()
    _run_child({"CABAC_TEST_SANITIZED_SHIM": so}, "test_host_shim.py")


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="needs the reference's sources (build container)")
def test_reference_adapter_under_asan_ubsan():
    """tests/test_reference_adapter.py — the reference's own CABACWriter / CABACReader call sequences through BinEncoderHipRef,
    BinDecoderHipRef, ResidualCoderHipRef and ResidualParserHipRef — against the sanitized build (the reference's translation
    units are compiled with the sanitizers too, where they lie)."""
    subprocess.check_call(["make", "-C", os.path.join(H.ROOT, "oracle"), "-j8", "_ref/libadapter_test_san.so"], stdout=subprocess.DEVNULL)
    so = os.path.abspath(os.path.join(H.ROOT, "oracle", "_ref", "libadapter_test_san.so"))
    _run_child({"CABAC_TEST_SANITIZED_ADAPTER": so}, "test_reference_adapter.py")
