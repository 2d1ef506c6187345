"""
bench.py — env-steps/s of the batched LocoEnv.step() hot path (BASELINE.json metric) on N GPUs of one node.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs-per-gpu 4096] [--task UnitreeA1.simple]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): UnitreeA1.simple, 4096 environments per GPU,
zero action, initial states = the 300 samples of the bundled mini dataset drawn with RandomState(0),
device-side auto-reset on _has_fallen or after 1000 control steps. A "step" = one control step of every
environment (= 10 physics substeps + observation + reward + termination + resets), one kernel launch.
Environments are independent: ranks shard them (weak scaling), the only collective is the metric
all-reduce at report time: ncclAllReduce on librccl.so through ctypes (loco_mujoco_amd/utils/collective.py).

`--task` switches to the other BASELINE robots for side measurements (HumanoidTorque.run / Atlas.walk /
HumanoidMuscle.run with the device's random policy a ~ U(-1,1)); the driver's default run is the A1 line above.

The timed region holds inputs resident in HBM (state lives on the device); it is bracketed by a barrier +
device synchronisation on both sides, the maximum over ranks is taken.
roofline: this path is not HBM-bound; `achieved` = algorithmic bytes per launch
(636 B per env-step incl. warm start x envs) / mean kernel duration measured with HIP events on the
library's own stream (lm_rollout returns it), against the 8 TB/s HBM peak — expect ~1e-4..1e-3.
cpu_baseline: the fp64 C oracle restatement ("port"), single thread, on a bounded sample of the same
workload (rank 0, N=1 only).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def algorithmic_bytes_per_env_step(nq, nv, nu, nobs, na=0):
    # state in + state out + action + obs + reward + done, plus the solver warm start in/out and the muscle
    # activations in/out
    return 4 * (2 * nq + 2 * nv + nu + nobs + 2) + 8 * nv + 8 * na


def parity_sample(env, hm, table, random_policy, n=64):
    """Second half of the metric: qpos / qvel L-infinity of the device against the fp64 oracle port after one control step
    from n dataset states under the workload's policy (checker only, outside the timed region)."""
    from loco_mujoco_amd.backend import HipBatch
    from oracle.model_blob import pack_model
    from oracle.pyoracle import Oracle
    m = env._model
    nv, na = m.nv, getattr(m, "na", 0)
    oracle = Oracle(pack_model(m))
    rs = np.random.RandomState(5)
    rows = table[rs.randint(0, len(table), n)]
    nu = len(env._action_indices)
    acts = rs.uniform(-1, 1, (n, nu)) if random_policy else np.zeros((n, nu))
    b = HipBatch(hm, n)
    b.set_state(rows[:, :nv], rows[:, nv:2 * nv])
    if rows.shape[1] > 2 * nv:
        b.set_goal(rows[:, 2 * nv:])
    b.step(acts)
    q, v = b.get_state()
    eq = ev = 0.0
    used = 0
    for i in range(n):
        ctrl = np.zeros(m.nu)
        ctrl[env._action_indices] = env._preprocess_action(acts[i])
        q0, v0 = rows[i, :nv].astype(np.float32).astype(np.float64), rows[i, nv:2 * nv].astype(np.float32).astype(np.float64)
        if na:
            qo, vo, _, _, st = oracle.step_act(q0, v0, np.zeros(na), ctrl, 10)
        else:
            qo, vo, _, st = oracle.step(q0, v0, ctrl, 10)
        if st["unhandled_pairs"]:
            continue                                         # a collider-less geom within reach of the floor on either side
        used += 1
        eq, ev = max(eq, np.abs(q[i] - qo).max()), max(ev, np.abs(v[i] - vo).max())
    tol = dict(qpos=1e-4, qvel=1e-2)
    return dict(qpos_linf=eq, qvel_linf=ev, states=used, against="fp64 oracle port (CPU), one control step = 10 substeps, "
                "same (qpos, qvel, ctrl)", tolerance=tol, within_tolerance=bool(used > 0 and eq <= tol["qpos"] and ev <= tol["qvel"]))


def leg_rate(envs_per_gpu, world, steps, seconds):
    """env-steps/s of ONE timed block: every environment of every rank advances `steps` control steps in `seconds` (the maximum
    over the ranks). The only place a rate is computed — cumulative device counters never enter a rate (round 3 divided the
    cumulative count of all legs by the last leg's time)."""
    return dict(value=envs_per_gpu * world * steps / seconds, ms_per_step=1e3 * seconds / steps)


def baseline_metric():
    """The headline metric, spelled exactly as /root/repo/BASELINE.json spells it."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "env-steps/sec at 4096 envs/GPU; qpos L\u221e vs CPU MuJoCo"


def cpu_baseline_all_cores(task, random_policy, make_kw, budget_s=8.0):
    """The same loop in one process per host core (fresh interpreters without torch / HIP), counts summed."""
    import subprocess
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                     # a container's CPU quota, not the host's core count, is what can run at once
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--task", task, "--cpu-budget", str(budget_s)]
    if make_kw:
        cmd.append("--dr")
    if random_policy:
        cmd.append("--cpu-random-policy")
    t0 = time.perf_counter()
    procs = [subprocess.Popen(cmd + ["--cpu-seed", str(i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for i in range(cores)]
    outs = [p.communicate()[0].decode().strip().splitlines() for p in procs]
    wall = time.perf_counter() - t0
    rates = [float(o[-1].split()[1]) for o in outs if o and o[-1].startswith("RATE")]
    if len(rates) < max(1, cores // 2):
        return None
    return dict(value=sum(rates), cores=len(rates), per_core=sum(rates) / len(rates), wall_s=wall)


def cpu_baseline(env, table, task, random_policy, budget_s=12.0, seed=0):
    """fp64 oracle restatement, one thread, same policy and initial-state distribution; bounded sample."""
    from oracle.model_blob import pack_model
    from oracle.pyoracle import Oracle
    m = env._model
    oracle = Oracle(pack_model(m))
    # timing: the same physics as the device simulates. The oracle's bookkeeping of geom pairs WITHOUT a collider (exact GJK
    # distances of convex hulls, test instrumentation that no simulator needs) stays out of the timed loop
    oracle.set_option("skip_pair_counter", 1)
    rs = np.random.RandomState(seed)
    nv, na = m.nv, getattr(m, "na", 0)
    qi = [m.jnt_id(n) for k, n, t in env.obs_helper.observation_spec[2:] if k.startswith("q_")]
    vi = [m.jnt_id(n) for k, n, t in env.obs_helper.observation_spec[2:] if k.startswith("dq_")]
    steps, t0, n_env = 0, time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s:
        row = table[rs.randint(0, len(table))]
        q, v, w, act = row[:nv].copy(), row[nv:2 * nv].copy(), np.zeros(nv), np.zeros(na)
        n_env += 1
        for _ in range(25):
            ctrl = np.zeros(m.nu)
            if random_policy:
                ctrl[env._action_indices] = env._preprocess_action(rs.uniform(-1, 1, len(env._action_indices)))
            if na:
                q, v, act, w, _ = oracle.step_act(q, v, act, ctrl, 10, w)
            else:
                q, v, w, _ = oracle.step(q, v, ctrl, 10, w)
            steps += 1
            if env._has_fallen(np.concatenate([q[qi], v[vi], row[2 * nv:]])):
                break
    dt = time.perf_counter() - t0
    return dict(value=steps / dt, unit="env-steps/s", cores=1, kind="port",
                sample="%d control steps over %d episodes of %s (%s, until fallen or 25 steps), "
                       "fp64 C oracle restatement, 1 thread, %.1f s"
                       % (steps, n_env, task, "random action" if random_policy else "zero action", dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--task", default="UnitreeA1.simple")
    ap.add_argument("--dr", action="store_true", help="Atlas.walk only: BASELINE config 4 — back joints kept, joint damping "
                    "redrawn per episode from the reference's domain_randomization_atlas.yaml")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)       # one process of the all-cores CPU baseline
    ap.add_argument("--cpu-budget", type=float, default=8.0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seed", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-random-policy", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--share-gpu", action="store_true", help="testing only: run all ranks on GPU 0 with the gloo backend "
                    "(checks the multi-rank control flow on a one-GPU box; the numbers mean nothing)")
    ap.add_argument("--dump-states", default=None, help=argparse.SUPPRESS)              # tests: <prefix>.rank<r>.npz with the final states
    ap.add_argument("--require-rccl", action="store_true", help="exit non-zero instead of reducing the metrics over TCP sockets when the RCCL "
                    "communicator does not come up (N > 1). The DEFAULT whenever WORLD_SIZE > 1 and every rank has a GPU of its own")
    ap.add_argument("--allow-tcp-fallback", action="store_true", help="N > 1: let the metric reduction fall back to the rendezvous sockets "
                    "when the RCCL communicator does not come up (the bench line names the transport in config.collective)")
    ap.add_argument("--sustained", type=int, default=200, help="control steps of the extra sustained leg (per-step launches, one "
                    "timed block of at least this many steps; 0 = skip; skipped when --steps already covers it)")
    ap.add_argument("--no-pollers", action="store_true", help="profiling runs (rocprofv3 runs one kernel at a time): the replay kernel only as the "
                    "pass behind the regular launch, no polling workgroups beside it (lm_batch_set_replay(3))")
    ap.add_argument("--fuse", type=int, default=25, help="control steps per launch of the extra fused-rollout leg (0/1 = skip)")
    args = ap.parse_args()

    if args.cpu_worker:
        from loco_mujoco_amd import LocoEnv
        np.random.seed(0)
        kw = {}
        if args.dr:
            import loco_mujoco_amd
            kw = dict(disable_back_joint=False)
        env = LocoEnv.make(args.task, debug=True, **kw)
        r = cpu_baseline(env, env._reset_table(), args.task, args.cpu_random_policy, args.cpu_budget, seed=args.cpu_seed)
        print("RATE %.3f" % r["value"])
        return

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    # the only collective of the path: the metric all-reduce at report time — ncclAllReduce on librccl.so through ctypes
    # (loco_mujoco_amd/utils/collective.py; no PyTorch anywhere in this file). --share-gpu: all ranks on GPU 0 and the
    # reduction over the rendezvous sockets (RCCL refuses two ranks on one device)
    from loco_mujoco_amd.utils.collective import Collective, MAX, SUM
    if args.share_gpu:
        local_rank = 0
    from loco_mujoco_amd import backend as _be
    # one rank per GPU on a node that has a GPU for every rank: the reduction IS RCCL or the run fails (no silent socket fallback)
    own_gpu = world > 1 and not args.share_gpu and _be.load_library().lm_device_count() >= world
    require_rccl = (args.require_rccl or (own_gpu and not args.allow_tcp_fallback)) and not args.share_gpu
    coll = Collective(backend="tcp" if args.share_gpu else "rccl", rank=rank, world=world, device=local_rank,
                      require_rccl=require_rccl)

    from loco_mujoco_amd import LocoEnv
    from loco_mujoco_amd.backend import HipBatch, HipModel

    n = args.envs_per_gpu
    default_task = args.task == "UnitreeA1.simple"
    action_mode = 0 if default_task else 1            # zero action (config 2) | device random policy (configs 3-5)
    np.random.seed(0)
    make_kw = {}
    if args.dr:
        assert args.task == "Atlas.walk", "--dr is BASELINE config 4 (Atlas.walk)"
        import loco_mujoco_amd
        make_kw = dict(disable_back_joint=False, domain_randomization_config=os.path.join(
            os.path.dirname(loco_mujoco_amd.__file__), "environments", "data", "atlas", "domain_randomization_atlas.yaml"))
    env = LocoEnv.make(args.task, debug=True, **make_kw)
    table = env._reset_table()
    hm = HipModel(env._chain_model(), device=local_rank)
    b = HipBatch(hm, n)
    if args.no_pollers:
        b.set_replay(3)
    offset = rank * n
    nv = env._model.nv
    rs = np.random.RandomState(0)
    if default_task:
        traj, step = rs.randint(0, 3, n * world), rs.randint(0, 100, n * world)
        pick = traj * 100 + step
    else:
        pick = rs.randint(0, len(table), n * world)
    rows = table[pick[offset:offset + n]]
    b.set_reset_table(table, seed=0, global_env_offset=offset)
    b.set_auto_reset(True, horizon=env.info.horizon)
    b.set_state(rows[:, :nv], rows[:, nv:2 * nv])
    if rows.shape[1] > 2 * nv:
        b.set_goal(rows[:, 2 * nv:])
    if args.dr:
        d = env._domain_rand.sample(n)                       # first episode: host draw; restarts: device redraw
        b.set_dof_params(damping=d[0], stiffness=d[1], frictionloss=d[2])
        b.set_dof_randomization(env._domain_rand.spec)

    def barrier():
        b.sync()                       # device synchronisation of this rank's stream ...
        coll.barrier()                 # ... then every rank has arrived (no-op for one rank)
        b.sync()

    def timed_leg(n_steps, seed, steps_per_launch=1):
        """One timed block: barrier + device sync, this rank's clock around exactly `n_steps` control steps + device sync, barrier.
        The clock stops BEFORE the closing barrier (a host-staged all-reduce): the maximum over the ranks of these per-rank times,
        taken at report time, is what a clock around both barriers would show less the collective's own latency. The device
        counters are reset first, so the returned statistics are this block's alone."""
        b.stats(reset=True)
        barrier()
        t = time.perf_counter()
        stats = b.rollout(n_steps, action_mode=action_mode, seed=seed, steps_per_launch=steps_per_launch)
        b.sync()
        dt = time.perf_counter() - t
        barrier()
        return dt, stats

    b.rollout(args.warmup, action_mode=action_mode, seed=11)
    elapsed, st = timed_leg(args.steps, 12)

    # extra leg, reported beside `value`, never as `value`: the same number of control steps with --fuse steps per launch
    # (lm_rollout_fused: no device-wide join between control steps; bitwise the same results). Continues from the state
    # the timed region left, so it runs the same mixture of walking and collapsing robots.
    fused = None
    if args.fuse > 1:
        b.rollout(args.fuse, action_mode=action_mode, seed=13, steps_per_launch=args.fuse)
        dtf, stf = timed_leg(args.steps, 14, steps_per_launch=args.fuse)
        fused = [dtf, stf["kernel_ms"]]

    # second extra leg: a sustained block of per-step launches (thermal / clock steady state rather than a short burst);
    # the timed region above already is one when --steps >= --sustained
    sustained = None
    if args.sustained > args.steps:
        dts, sts = timed_leg(args.sustained, 15)
        sustained = [dts, sts["env_steps"]]

    vals = np.array([elapsed, st["env_steps"], st["episodes"], st["reward_sum"], st["nan_resets"],
                     st["overflow_contacts"], st["unhandled_geoms"], st["solver_iters"], st["kernel_ms"],
                     fused[0] if fused is not None else 0.0, st["self_proximity"], st["self_contacts"],
                     sustained[0] if sustained is not None else 0.0, sustained[1] if sustained is not None else 0.0,
                     st.get("replayed_env_steps", 0.0)], dtype=np.float64)
    tmax = coll.all_reduce(vals, MAX)
    vals = coll.all_reduce(vals, SUM)
    elapsed, kernel_ms, fused_elapsed = float(tmax[0]), float(tmax[8]), float(tmax[9])
    if args.dump_states:                                      # tests: this rank's final states, to compare partitions
        q_fin, v_fin = b.get_state()
        np.savez(args.dump_states + ".rank%d.npz" % rank, qpos=q_fin, qvel=v_fin, offset=offset)
    coll.close()                                              # every collective happens before the non-zero ranks leave
    if rank != 0:
        return
    env_steps = vals[1]
    main_leg = leg_rate(n, world, args.steps, elapsed)
    value = main_leg["value"]
    # the device's own count of this block (counters reset right before it) must be the same number of env-steps
    assert abs(env_steps - n * world * args.steps) < 0.5, (env_steps, n, world, args.steps)
    m = env._model
    forwards = 40 if m.integrator else 10             # RK4: four forward passes per substep
    per_env_step = algorithmic_bytes_per_env_step(nv, nv, len(env._action_indices), b.nobs, getattr(m, "na", 0))
    bytes_per_launch = per_env_step * n
    launch_s = kernel_ms * 1e-3 / args.steps
    achieved = bytes_per_launch / launch_s / 1e9
    traffic = None
    valu = None
    # counters of the committed profile (tools/probes/prof_run.sh: the same command under rocprofv3, separate --pmc passes).
    # They are only quoted when the profile was taken on THIS build of the library (sha256 of liblocohip.so stamped into it):
    # after a kernel change they go stale, and stale numbers are dropped (null) rather than reported.
    import hashlib
    from loco_mujoco_amd import backend as _backend
    lib_sha = hashlib.sha256(open(_backend.LIB_PATH, "rb").read()).hexdigest()[:16]
    # profiles/<tag>_pmc.json: "r4" for the bench line, "r4_<task>[.dr][<envs>]" for the other configurations
    tag = "r4" if (default_task and n == 4096) else "r4_%s%s%s" % (args.task, ".dr" if args.dr else "", "" if n == 4096 else str(n))
    prof = os.path.join(ROOT, "profiles", tag + "_pmc.json")
    prof_name = "profiles/%s_pmc.json" % tag
    prof_note = "no committed profile for this workload"
    if os.path.exists(prof) and json.load(open(prof)).get("lib_sha16") != lib_sha:
        prof_note = "%s was taken on another build of liblocohip.so (%s, this one is %s): counters not quoted" % (prof_name, json.load(open(prof)).get("lib_sha16"), lib_sha)
    elif os.path.exists(prof):
        prof_note = "%s, taken on this build (liblocohip.so sha256[:16] = %s)" % (prof_name, lib_sha)
        try:
            pmc = json.load(open(prof))["pmc"]
            # separate --pmc passes (tools/probes/prof_run.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950)
            traffic = pmc["FETCH_SIZE"]["bytes_per_dispatch_corrected_x2"] + pmc["WRITE_SIZE"]["bytes_per_dispatch"]
            # what actually bounds the kernel: VALU issue. One wave64 VALU instruction holds a 16-lane SIMD for 4 cycles;
            # SQ_WAVE_CYCLES counts in units of 4 cycles (it reproduces the mean wave time measured with s_memtime).
            prof_ns = json.load(open(prof))["duration_ns"]["avg"]
            valu = dict(valu_insts_per_wave=pmc["SQ_INSTS_VALU"]["per_dispatch"] / pmc["SQ_WAVES"]["per_dispatch"],
                        valu_busy_frac_of_wave_time=pmc["SQ_INSTS_VALU"]["per_dispatch"] / pmc["SQ_WAVE_CYCLES"]["per_dispatch"],
                        mean_wave_time_over_launch_time=4.0 * pmc["SQ_WAVE_CYCLES"]["per_dispatch"] / pmc["SQ_WAVES"]["per_dispatch"]
                        / (prof_ns * 2.4),
                        note=prof_name + "; 2.4 GHz assumed; with one wave per SIMD (4096 environments) the SIMD's VALU issue rate "
                             "is the product of the two fractions")
        except Exception:
            traffic = None
    out = {
        "metric": baseline_metric(),
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": main_leg["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s, %d envs/GPU, %s rollout, device-side auto-reset "
                               "(horizon 1000), 10 physics substeps per env-step"
                               % (args.task + (" (back joints, joint-damping randomisation per episode)" if args.dr else ""), n,
                                  "zero-action" if default_task else "random-policy"),
                   "envs_per_gpu": n, "global_envs": n * world, "parallelism": "env-sharded x%d" % world,
                   "collective": {"none": "none (one rank)", "rccl": "ncclAllReduce (RCCL) of 15 doubles at report time",
                                  "tcp": "socket reduction of 15 doubles at report time (RCCL not used)"}[coll.backend]},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "kernel": "step_kernel<3 links,6 slots,Euler,elliptic,self-collisions,4 replicas>" if default_task else "step_kernel",
                     "profile": prof_note, "lib_sha16": lib_sha,
                     "kernel_ms_per_launch": 1e3 * launch_s,
                     "algorithmic_bytes_per_env_step": per_env_step,
                     "note": "path is VALU-issue/latency-bound by design (SURVEY.md 8d): %d algorithmic B per env-step; "
                             "traffic = PMC bytes per launch from the committed profile (same command, separate rocprofv3 "
                             "--pmc passes), null when that profile is not of this build" % per_env_step},
        "stats": {"episodes": vals[2], "mean_reward": vals[3] / max(env_steps, 1), "nan_resets": vals[4],
                  "overflow_contacts": vals[5], "unhandled_geom_substeps": vals[6],
                  # where the device left its validated collision model (all ranks): forward passes x geom pairs of the robot
                  # without a pair collider (box / cylinder) within the margin, and self-contacts it did simulate
                  "self_proximity": vals[10], "self_contacts": vals[11],
                  # control steps that left the regular kernel's capacity (contact slots, pair lists) and were run by the replay kernel
                  "replayed_env_steps": vals[14],
                  "newton_iters_per_forward_pass": vals[7] / max(env_steps * forwards, 1),
                  "physics_substeps_per_s": 10 * value},
    }
    if valu is not None:
        out["roofline"]["valu"] = valu
    if fused is not None:
        fl = leg_rate(n, world, args.steps, fused_elapsed)
        out["rollout_fused"] = {"steps_per_launch": args.fuse, "value": fl["value"], "unit": "env-steps/s",
                                "ms_per_step": fl["ms_per_step"],
                                "note": "policy-free rollout with %d control steps per launch (lm_rollout_fused): every "
                                        "environment advances on its own, results bitwise those of single-step launches; "
                                        "a policy in the loop gets `value`" % args.fuse}
    if sustained is not None and tmax[12] > 0:
        sl = leg_rate(n, world, args.sustained, float(tmax[12]))
        assert abs(vals[13] - n * world * args.sustained) < 0.5, (vals[13], n, world, args.sustained)    # this block's own count
        out["sustained"] = {"steps": args.sustained, "value": sl["value"], "unit": "env-steps/s",
                            "ms_per_step": sl["ms_per_step"],
                            "note": "one timed block of %d per-step launches after the timed region (same policy, same "
                                    "state mixture); `value` is the --steps block" % args.sustained}
    elif args.sustained:
        out["sustained"] = {"steps": args.steps, "value": value, "unit": "env-steps/s", "ms_per_step": 1e3 * elapsed / args.steps,
                            "note": "the timed region itself (--steps >= --sustained)"}
    if world == 1 and not args.no_cpu_baseline:
        out["parity"] = parity_sample(env, hm, table, not default_task)
        one = cpu_baseline(env, table, args.task, not default_task, budget_s=6.0)
        allc = cpu_baseline_all_cores(args.task, not default_task, make_kw)
        # reported baseline = every host core running the fp64 port (falls back to the single-core sample)
        out["cpu_baseline"] = dict(one)
        out["cpu_baseline"]["single_core"] = one["value"]
        if allc is not None:
            out["cpu_baseline"].update(value=allc["value"], cores=allc["cores"],
                                       sample=one["sample"] + "; value = the same loop in %d processes (one per host core) for "
                                       "%.0f s each, counts summed (%.0f env-steps/s per process under full load; process count = CPU affinity "
                                       "capped by the cgroup quota)"
                                       % (allc["cores"], 8.0, allc["per_core"]))
        out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    print(json.dumps(out))
    if "parity" in out and not out["parity"]["within_tolerance"]:
        # the second half of the metric failed: the line above says so ("within_tolerance": false), and so does the exit code
        print("bench: device vs fp64 oracle beyond the stated tolerance: qpos %.3g (tol %.0e), qvel %.3g (tol %.0e)"
              % (out["parity"]["qpos_linf"], out["parity"]["tolerance"]["qpos"], out["parity"]["qvel_linf"],
                 out["parity"]["tolerance"]["qvel"]), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
