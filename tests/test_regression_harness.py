"""integration/CMakeLists.txt + run_test_gpu.cmake: the GPU twin of the reference's ctest (CMakeLists.txt:55-101,
test/run_test.cmake, test/hashes.txt).  The VTM fork and the y4m clips do not exist offline, so what can be checked is that
the project configures, says why it cannot run, and registers a ctest entry that is reported as SKIPPED (not passed)."""
import os
import shutil
import subprocess

import pytest

import helpers as H


@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not installed")
def test_harness_configures_and_reports_skipped(tmp_path):
    args = ["cmake", "-S", os.path.join(H.ROOT, "integration"), "-B", str(tmp_path)]
    if os.path.isdir("/root/reference"):
        args.append("-DREFERENCE_DIR=/root/reference")
    r = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "skipped: VTM/clips not supplied" in r.stdout
    t = subprocess.run(["ctest"], cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert t.returncode == 0 and "gpu_regression_skipped" in t.stdout and "Skipped" in t.stdout, t.stdout + t.stderr
    assert "Passed" not in t.stdout.split("gpu_regression_skipped")[1].splitlines()[0]


def test_runner_script_skips_without_the_application(tmp_path):
    r = subprocess.run(["cmake", "-DCMD=/nonexistent/EncoderApp", "-DARGS=-i x", "-DLOG_FILE=bin_log.txt", "-DOUT_FILE=str.bin",
                        "-P", os.path.join(H.ROOT, "integration", "run_test_gpu.cmake")], capture_output=True, text=True,
                       cwd=str(tmp_path), timeout=120)
    assert r.returncode == 0 and "skipped: VTM/clips not supplied" in (r.stdout + r.stderr)
