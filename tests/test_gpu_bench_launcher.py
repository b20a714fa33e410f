"""bench.py under the launcher the driver uses (one rank here: the box has one GPU): the RCCL process group is created before any other
GPU call, the timed region is bracketed by barriers, rank 0 prints ONE JSON line with the contract keys."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_under_torch_distributed_run(gpu):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SDENG_BENCH_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--spinup", "0.05"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "log_z_abs_err"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["scaling"] == "weak" and j["value"] > 1e8
    # the headline bound is the measured vector-port share when profiles/ holds counters of THIS library build (digest match), else the live
    # matrix-pipe utilisation; either way a fraction in (0, 1]
    assert j["roofline"]["bound"] in ("valu_issue", "mfma") and 0.0 < j["roofline"]["frac"] <= 1.0 and j["log_z_abs_err"] < 1e-3
    if j["roofline"]["bound"] == "valu_issue":
        assert j["roofline"]["traffic"] is not None and "library_digest" in j
