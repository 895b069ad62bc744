"""The multi-GPU path of bench.py through RCCL itself — as far as a one-GPU box allows: a process group of ONE rank
(CABAC_BENCH_DIST_ONE=1), backend "nccl", so that init_process_group with device_id, the barriers, the all_reduce / all_gather
calls on device tensors and the scatter / gather of sharding.py run on the GPU box's RCCL instead of only on gloo
(tests/test_sharding_gloo.py covers world sizes 2 and 3 on the CPU; N > 1 on GPUs is the driver's to run)."""
import json
import os
import subprocess
import sys

import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["weak", "strong"])
def test_bench_through_rccl_with_one_rank(mode):
    env = dict(os.environ, CABAC_BENCH_DIST_ONE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29541 + (mode == "strong")),
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cmd = [sys.executable, os.path.join(H.ROOT, "bench.py"), "--gpus", "1", "--workload", "C3", "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline", "--no-end-to-end", "--no-co-scheduled", "--no-residual"]
    if mode == "strong":
        cmd.append("--strong")
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["hash_match"] and line["hash_whole_batch"]["differ"] == 0 and line["n_gpus"] == 1
    assert line["scaling"] == mode
    if mode == "strong":
        assert line["strong"]["scatter_ms"] is not None and line["strong"]["gathered_payload_bytes"] > 0
    else:
        assert line["sizes_allgather_ms"] is not None
