"""the reference's own CABACWriter walking several substreams on the recording GPU encoder produces
the same bin_log.txt — the file whose md5 the reference's ctest pins (test/hashes.txt, test/run_test.cmake:9,
CMakeLists.txt:55-101) — and the same bytes as on BinEncoder_Std.  The reference is built with its ENABLE_LOGGING option
by oracle/Makefile (oracle/_ref/libcabac_ref_log.so, test infrastructure); the expected md5s of the walk are committed in
tests/golden/bin_log_walk.json (generated from the reference by oracle/gen_golden.py)."""
import json
import os
import subprocess
import sys

import pytest

import helpers as H

LOG_LIB = os.path.join(H.ORACLE_DIR, "_ref", "libadapter_test_log.so")
needs_log_build = pytest.mark.skipif(not os.path.exists(LOG_LIB), reason="ENABLE_LOGGING build of the reference not present")


def _run(mode, tmp_path):
    H.ref_test_library("libadapter_test_log.so")      # rebuilt here if stale (build container), refused on the GPU box
    out = str(tmp_path / "walk.json")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "bin_log_walk.py"), mode, out],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.load(open(out))


def _check(res, names):
    gold = json.load(open(os.path.join(H.GOLDEN, "bin_log_walk.json")))
    std = res["std"]
    assert std["log_lines"] > 2000 and std["log_bytes"] > 10000          # a real log: one line per syntax element
    assert std["log_md5"] == gold["log_md5"] and std["stream_md5"] == gold["stream_md5"]
    for n in names:
        assert res[n]["log_md5"] == std["log_md5"] and res[n]["log_bytes"] == std["log_bytes"], n
        assert res[n]["stream_md5"] == std["stream_md5"], n


@needs_log_build
def test_recording_front_keeps_bin_log_and_bytes(tmp_path):
    """CPU: CABACWriter on BinEncoderHipRef (recording, no device) writes the identical bin_log.txt, and its recorded
    bins, coded by the oracle, are the bytes BinEncoder_Std produced."""
    _check(_run("cpu", tmp_path), ["recorded"])


@needs_log_build
@pytest.mark.gpu
def test_device_path_keeps_bin_log_and_bytes(tmp_path):
    """GPU: the same walk with every substream coded by one HipBatch::flush() on the device."""
    _check(_run("gpu", tmp_path), ["recorded", "device"])
