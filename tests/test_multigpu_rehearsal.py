"""The N-rank path of bench.py end to end on ONE GPU (``MR_BENCH_REHEARSE=1``: every rank on device 0, gloo
instead of RCCL): launch under torch.distributed.run as the driver does, every rank renders its stripes or
band with the HIP library, the collective assembles the frame, rank 0 compares it with a single-device render
(an assert inside bench.py) and prints the JSON line.  Rates measured this way mean nothing and are not looked at."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ranks,config,partition", [(2, "c3", "stripes"), (3, "c2", "bands"), (3, "c3", "weighted")])
def test_bench_multi_rank_path_on_one_gpu(ranks, config, partition):
    env = dict(os.environ, MR_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(29530 + ranks), os.path.join(ROOT, "bench.py"),
           "--gpus", str(ranks), "--config", config, "--partition", partition, "--steps", "1", "--warmup", "1",
           "--frames-per-step", "24", "--no-cpu-baseline"]
    run = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["name"] == config and partition in out["config"]["parallelism"]
