"""
N>1 path on CPU (world_size 2, gloo): the environment partition (global env ids, reset-table offsets) and the metric
all-reduce bench.py performs at report time. Environments are independent, so this is all the communication there is.
"""

import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, %r)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 8
    offset = rank * n
    # the same initial-state assignment bench.py uses: one global RandomState stream, sliced per rank
    rs = np.random.RandomState(0)
    traj, step = rs.randint(0, 3, n * world), rs.randint(0, 100, n * world)
    mine = (traj * 100 + step)[offset:offset + n]
    # metrics: [elapsed, env_steps, episodes]: MAX for time, SUM for counts (bench.py)
    vals = torch.tensor([1.0 + rank, 100.0 * n, 3.0 + rank], dtype=torch.float64)
    tmax = vals.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(vals, op=dist.ReduceOp.SUM)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine.tolist())
    if rank == 0:
        print(json.dumps(dict(elapsed=float(tmax[0]), env_steps=float(vals[1]), episodes=float(vals[2]), rows=gathered)))
    dist.destroy_process_group()
""") % ROOT


def test_partition_and_metric_reduction_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    import numpy as np
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["elapsed"] == 2.0 and res["env_steps"] == 1600.0 and res["episodes"] == 7.0
    rs = np.random.RandomState(0)
    traj, step = rs.randint(0, 3, 16), rs.randint(0, 100, 16)
    assert sum(res["rows"], []) == (traj * 100 + step).tolist()      # the union of the shards is the global assignment
