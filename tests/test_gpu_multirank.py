"""The multi-rank control flow of bench.py, rehearsed on the ONE GPU of the test box: `python bench.py --gpus 2` from a bare
environment must start its own launcher as a child process, both ranks must run the replica rollouts and the data-parallel
training leg (four gradient all-reduces per iteration over the flat arena) and rank 0 must print one JSON line.  RCCL refuses
two ranks on one device, so the rehearsal switches put both ranks on GPU 0 with the gloo backend (DVG_BENCH_SHARE_GPU=1,
DVG_BENCH_BACKEND=gloo): the numbers mean nothing and the line says so; the code path - self-launch, rendezvous, barriers,
max-over-ranks timing, ArenaReducer ranges, the guarded graphed leg - is the one the 8-GPU run takes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_rehearsal():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DVG_BENCH_SHARE_GPU="1", DVG_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--model", "dcgan", "--no-families", "--no-cpu-baseline", "--train-iters", "1",
                        "--train-graph-timeout", "120"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "rehearsal" in d
    assert d["value"] > 0 and d["config"]["rollouts_in_flight"] == 3
    t = d["train"]
    assert t["rccl_ranks"] == 2                       # from an actual all-reduce of ones over the process group
    assert t["eager"]["allreduces_per_iter"] == 4.0   # decoder / LSTM / GP range, encoder range, LSTM range, GP range
    assert t["eager"]["allreduce_MB_per_iter"] > 40
    assert "eager_no_allreduce" in t and "hipgraph" in t
