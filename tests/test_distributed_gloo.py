"""
N>1 path on CPU (world_size 2, gloo): the environment partition (global env ids, reset-table offsets) and the metric
all-reduce bench.py performs at report time. Environments are independent, so this is all the communication there is.
"""

import os
import socket

import pytest
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


# ---------------------------------------------------------------------------------------------------------------------
# The product's own collective (loco_mujoco_amd/utils/collective.py; bench.py uses it, no PyTorch): world size 2 on the CPU
# through the launcher the driver uses. The "tcp" backend carries the reduction over the rendezvous sockets; the "rccl"
# backend needs GPUs and is exercised by the driver's multi-GPU runs.
# ---------------------------------------------------------------------------------------------------------------------
WORKER2 = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path.insert(0, %r)
    from loco_mujoco_amd.utils.collective import Collective, MAX, SUM
    coll = Collective(backend=os.environ.get("LM_TEST_COLLECTIVE", "tcp"))
    rank, world = coll.rank, coll.world
    vals = np.array([1.0 + rank, 800.0, 3.0 + rank, 0.25 * (rank + 1)])
    tmax = coll.all_reduce(vals, MAX)
    tsum = coll.all_reduce(vals, SUM)
    coll.barrier()
    coll.close()
    if rank == 0:
        print(json.dumps(dict(world=world, tmax=tmax.tolist(), tsum=tsum.tolist(), backend=coll.backend)))
""") % ROOT


@pytest.mark.parametrize("backend", ["tcp", "rccl"])
def test_product_collective_world2_through_the_launcher(tmp_path, backend):
    """"rccl" on a box without GPUs: the communicator cannot come up, both ranks agree (over the rendezvous sockets) to reduce
    over those sockets instead — the fallback of the driver's multi-GPU run should RCCL not initialise on its node."""
    script = tmp_path / "worker2.py"
    script.write_text(WORKER2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=300, env=dict(os.environ, LM_TEST_COLLECTIVE=backend))
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    # (on a node with two GPUs and a working RCCL the "rccl" request is served by RCCL: either is correct here)
    assert res["world"] == 2 and res["backend"] in (("tcp",) if backend == "tcp" else ("tcp", "rccl"))
    assert res["tmax"] == [2.0, 800.0, 4.0, 0.5] and res["tsum"] == [3.0, 1600.0, 7.0, 0.75]


def test_require_rccl_fails_loudly_without_gpus(tmp_path):
    """`bench.py --require-rccl` / Collective(require_rccl=True): when the communicator cannot come up, every rank raises instead
    of quietly reducing over the sockets (here: a box without GPUs)."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs: RCCL comes up")
    script = tmp_path / "worker3.py"
    script.write_text(textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        from loco_mujoco_amd.utils.collective import Collective
        try:
            Collective(backend="rccl", require_rccl=True)
        except RuntimeError as e:
            print("RAISED", e)
            sys.exit(3)
    """) % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=300)
    # (the two ranks' prints interleave on the shared pipe: match the message, not the prefix before it)
    assert out.returncode != 0 and "RCCL was required" in out.stdout + out.stderr


def test_bench_gpus_flag_makes_the_ranks_itself():
    """`python bench.py --gpus 2` WITHOUT a launcher starts two ranks itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on
    127.0.0.1) and reports n_gpus = 2 (`--plumbing-only`: ranks, rendezvous and the reduction, no GPU work); under a launcher whose
    WORLD_SIZE disagrees with --gpus it refuses; and on a node without a GPU per rank it fails loudly instead of reporting one rank."""
    import json
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, bench, "--gpus", "2", "--plumbing-only"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["ranks_counted"] == 2 and res["rank_sum"] == 1.0
    out = subprocess.run([sys.executable, bench, "--gpus", "4", "--plumbing-only"], capture_output=True, text=True, timeout=300,
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert out.returncode == 2 and "WORLD_SIZE=2" in out.stderr
    import torch
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")], out.stdout[-500:]


@pytest.mark.gpu
def test_rccl_binding_single_rank_communicator_on_the_gpu():
    """The ctypes binding itself on hardware (the GPU box has ONE device, where RCCL admits a one-rank communicator):
    ncclGetUniqueId, ncclCommInitRank with the 128-byte id BY VALUE, hipMalloc / hipMemcpy through the runtime liblocohip.so
    mapped, ncclAllReduce for SUM and MAX, ncclCommDestroy."""
    import ctypes as C
    import numpy as np
    from loco_mujoco_amd import backend as _b   # noqa: F401  (maps liblocohip.so and with it the HIP runtime)
    from loco_mujoco_amd.utils.collective import Collective, MAX, SUM
    c = Collective.__new__(Collective)
    c.rank, c.world, c.backend, c._peers, c._server, c._comm = 0, 1, "rccl", [], None, None
    c._load_libraries()
    c._check_hip(c._hip.hipSetDevice(0), "hipSetDevice")
    uid = (C.c_byte * 128)()
    c._check(c._nccl.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
    c._init_rccl(uid)
    v = np.array([1.5, -2.0, 800.0, 0.25])
    assert np.array_equal(c._reduce_rccl(v, SUM), v) and np.array_equal(c._reduce_rccl(v, MAX), v)
    assert c.comm_count() == 1                       # what bench.py reports as config.ranks_seen_by_rccl
    c.close()
    assert c._comm is None


@pytest.mark.gpu
def test_product_collective_world2_rccl_on_two_gpus(tmp_path):
    """Two ranks, two devices, `require_rccl`: the reduction bench.py reports with must be RCCL's (skips on the 1-GPU box)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs of one node")
    script = tmp_path / "worker4.py"
    script.write_text(WORKER2.replace('Collective(backend=os.environ.get("LM_TEST_COLLECTIVE", "tcp"))',
                                      'Collective(backend="rccl", require_rccl=True)'))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["world"] == 2 and res["backend"] == "rccl"
    assert res["tmax"] == [2.0, 800.0, 4.0, 0.5] and res["tsum"] == [3.0, 1600.0, 7.0, 0.75]
