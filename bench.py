#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched MultiCarRacing-v0 step on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (collide -> dynamics -> view raster, incl. device-side auto-reset and
host-side episode generation for finished envs) over one batch of 4096 envs per GPU with synthetic random
actions already resident in HBM.  Workload at N=1 = BASELINE.json configs[1]: num_agents=2, batch=4096,
96x96 RGB obs.  Multi-GPU: independent env slices (weak scaling), one scalar all-reduce at the end.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (view/raster kernel vs HBM) and `cpu_baseline`
(the oracle = CPU restatement, timed on this host's cores on a bounded sample; N=1 only).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cpu_leg(O, envs, num_agents, obs, threads, seed, t0_step, budget_s):
    """Time the oracle on `envs` with the counter-based synthetic action stream (host twin of what the GPU consumes)
    for about `budget_s` seconds.  Returns (env-steps/s, steps, seconds, next step index)."""
    import ctypes
    import numpy as np
    from multi_car_racing_amd import _lib
    L = _lib.load()
    n = len(envs)

    def acts(t0, k):
        a = np.zeros((k, n, num_agents, 3), np.float32)
        for i in range(k):
            L.mcr_synth_actions_host(_lib.ptr(a[i]), n, num_agents, ctypes.c_uint64(seed), ctypes.c_uint32(t0 + i), ctypes.c_uint32(0))
        return a
    probe = 20
    a = acts(t0_step, probe)
    t0 = time.perf_counter(); O.rollout(envs, a, render=obs, threads=threads); dt = time.perf_counter() - t0
    t0_step += probe
    steps = int(max(probe, min(4000, probe * budget_s / max(dt, 1e-4))))
    a = acts(t0_step, steps)
    t0 = time.perf_counter(); O.rollout(envs, a, render=obs, threads=threads); dt = time.perf_counter() - t0
    return n * steps / dt, steps, dt, t0_step + steps


def cpu_baseline(num_agents, obs):
    """Oracle (C++ CPU restatement of the same step, oracle/mcr_oracle.cpp) on a bounded sample of the same workload:
    8 envs per host core, same counter-based action stream, one OpenMP thread per core, no Python inside the timed
    loops.  Headline leg: all cores, obs as benched; extra legs: 1 thread, and obs off.  ~25 s in total."""
    from oracle import oracle as O
    from tests.util import oracle_episode
    from multi_car_racing_amd._lib import effective_cpus
    cores = effective_cpus()                   # min(affinity, cgroup CPU quota): what this job may really use
    n_envs = 8 * cores
    envs = []
    for e in range(n_envs):
        o = O.OracleEnv(num_agents)
        o.reset(oracle_episode(O, num_agents, 12345, e, use_random_direction=True), render=False)
        envs.append(o)
    seed = 4321
    _, _, _, t = _cpu_leg(O, envs, num_agents, obs, cores, seed, 0, 1.0)          # warm-up through the zoom-in (>= 50 steps)
    v, steps, dt, t = _cpu_leg(O, envs, num_agents, obs, cores, seed, t, 10.0)
    v1, s1, d1, t = _cpu_leg(O, envs[:8], num_agents, obs, 1, seed, t, 5.0)
    v0, s0, d0, t = _cpu_leg(O, envs, num_agents, False, cores, seed, t, 5.0)
    for o in envs:
        o.close()
    # BASELINE configs[0]: the reference's own CPU-runnable case — ONE env, num_agents=1 (the CarRacing-v0 special case of README:66-71:
    # use_random_direction=False, backwards_flag=False), state_pixels every step, one thread
    o1 = O.OracleEnv(1, backwards_flag=False)
    o1.reset(oracle_episode(O, 1, 12345, 0, use_random_direction=False), render=False)
    _, _, _, tc = _cpu_leg(O, [o1], 1, True, 1, seed, 0, 0.5)
    vc, sc, dc, _ = _cpu_leg(O, [o1], 1, True, 1, seed, tc, 3.0)
    o1.close()
    what = "oracle/mcr_oracle.cpp with OpenMP over envs (CPU restatement; the reference's Box2D+pyglet path is not installable here)"
    return {"value": v, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n_envs} envs x {steps} steps, num_agents={num_agents}, obs={'96x96x3' if obs else 'none'}, {what}, "
                      f"{dt:.1f} s wall; host reports {os.cpu_count()} logical CPUs, cgroup/affinity allows {cores}",
            "legs": [{"value": v1, "unit": "env-steps/s", "cores": 1, "sample": f"8 envs x {s1} steps, obs={'96x96x3' if obs else 'none'}, {d1:.1f} s"},
                     {"value": v0, "unit": "env-steps/s", "cores": cores, "sample": f"{n_envs} envs x {s0} steps, obs=none (physics + bookkeeping only), {d0:.1f} s"},
                     {"value": vc, "unit": "env-steps/s", "cores": 1, "sample": f"BASELINE configs[0]: 1 env x {sc} steps, num_agents=1, use_random_direction=False, backwards_flag=False, obs=96x96x3, {dc:.1f} s"}]}


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n:
        sys.stderr.write(f"bench.py: --gpus {n} requested but this node exposes {have} HIP device(s); refusing to report a "
                         f"{n}-GPU number from fewer GPUs\n")
        sys.exit(2)
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--agents", type=int, default=2)
    ap.add_argument("--obs", type=int, default=1)
    ap.add_argument("--streams", type=int, default=2, help="2 (default): contact side stream (envs in car<->car contact run their chain concurrently); 1: single stream")
    ap.add_argument("--stagger", type=int, default=1, help="1: spread the TimeLimit phases of the envs uniformly before timing (steady state); 0: all envs expire in the same step")
    ap.add_argument("--debug-bits", type=int, default=0, help="mcr_debug_set value for timing experiments (results are WRONG when non-zero)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--time-all-kernels", action="store_true", help="HIP-event time all three kernels (adds overhead)")
    ap.add_argument("--action-seed", type=int, default=1234)
    ap.add_argument("--actions", choices=["random", "drive"], default="random", help="random (default): i.i.d. actions over the action space (BASELINE's random-action rollout); "
                    "drive: gas 1, brake 0, steering noise +-0.1 from the same counter-based stream — cars that actually drive off the grid, lap and collide "
                    "(what a competent policy's rollout looks like to the contact chain); reported with the contact-list length")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket the raster launch with HIP events (no roofline object; lets --graph 1 replay)")
    ap.add_argument("--emulate-world", type=int, default=0, help="W > 1: ONE GPU, but the host side of a W-rank job on this node: the process is pinned to 1/W of the "
                    "cores the cgroup allows and its track generator takes the threads VecMultiCarRacing gives a rank of a W-rank job; reports "
                    "env-steps/s, the time step() was blocked on the refill thread and the env-steps frozen waiting for the host")
    ap.add_argument("--rccl", action="store_true", help="with --gpus 1: initialise a 1-rank RCCL (\"nccl\") process group anyway, so that the metric all-reduce of "
                    "sharded.reduce_metrics and the step's phase-word ordering run beside a live RCCL communicator on the one GPU a box has")
    ap.add_argument("--terminal-obs", type=int, default=0, help="1: also hand out the last frame of every episode that ends (info['terminal_observation'], include/mcr.h: mcr_set_terminal_obs)")
    ap.add_argument("--refill", choices=["native", "python"], default="native", help="native (default): the handle's own host thread polls, generates and stages the "
                    "consumed episodes (include/mcr.h: mcr_refill_start); python: rounds 2-5's worker thread in vec_env.py")
    ap.add_argument("--sync-every", type=int, default=0, help="S > 0: the stepping loop synchronises its stream after every S-th step (S = 1: an RL loop that reads its "
                    "observations before it picks the next actions) instead of running up to 16 steps ahead of the GPU; not the headline's loop")
    ap.add_argument("--fresh-world", type=int, default=0, help="0 (default): ONE b2World per env across its episodes, as the reference keeps it (csrc/k_world.h); "
                    "1: every episode the first of a fresh world (rounds 1-5's definition)")
    ap.add_argument("--graph", type=int, default=0, help="1: mcr_step replays a hipGraph of the step (bypassed while kernels are timed; measured gain 0.4 %%); 0 (default): plain launches")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        _spawn_ranks(args.gpus)                 # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks\n")
        sys.exit(2)

    emu_share = None
    if args.emulate_world > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1:
        # the host share of ONE rank of a W-rank job on this node: the ranks split the cores the cgroup allows (vec_env.py).  Pinned
        # BEFORE the HIP runtime starts, so that its own threads (signal handling) live on the share as well, like everything the rank runs.
        from multi_car_racing_amd._lib import effective_cpus      # (no HIP call in there: min(affinity, cgroup CPU quota))
        allowed = sorted(os.sched_getaffinity(0)); total = effective_cpus()
        emu_share = (total, max(1, total // args.emulate_world))
        os.sched_setaffinity(0, set(allowed[:emu_share[1]]))
    import ctypes
    import numpy as np
    import torch
    import torch.distributed as dist
    from multi_car_racing_amd.sharded import ShardedVecEnv, reduce_metrics

    def thread_cpu():
        """CPU seconds (user + system) of every thread of this process by name: who on the host side is busy"""
        out, tck = {}, os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            try:
                name = open(f"/proc/self/task/{tid}/comm").read().strip()
                f = open(f"/proc/self/task/{tid}/stat").read().rsplit(")", 1)[1].split()
                out[(tid, name)] = (int(f[11]) + int(f[12])) / tck
            except OSError:
                pass
        return out

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_rank >= torch.cuda.device_count():
        sys.stderr.write(f"bench.py: rank {rank} needs HIP device {local_rank}, the node exposes {torch.cuda.device_count()}\n")
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif args.rccl:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(_free_port()))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        warm = torch.ones(4, device=torch.device("cuda", local_rank)); dist.all_reduce(warm); torch.cuda.synchronize()    # the communicator and its streams exist
    dev = torch.device("cuda", local_rank)

    B, N, K, W = args.envs, args.agents, args.steps, args.warmup
    emu = None
    extra = {}
    if args.emulate_world > 1 and world == 1:
        # the host share of ONE rank of a W-rank job on this node: the ranks split the cores the cgroup allows (vec_env.py)
        total, share = emu_share
        extra["gen_threads"] = max(1, total // args.emulate_world)     # what VecMultiCarRacing gives a rank of a W-rank job
        emu = {"world": args.emulate_world, "cores_allowed_to_the_job": total, "cores_this_rank": share, "gen_threads": extra["gen_threads"]}
    env = ShardedVecEnv(B * world, N, seed=0, rank=rank, world_size=world, device=dev, obs=bool(args.obs),
                        auto_reset=True, use_random_direction=True, streams=args.streams, graph=bool(args.graph), terminal_obs=bool(args.terminal_obs), fresh_world=bool(args.fresh_world), async_refill=(True if args.refill == "native" else "python"), **extra)
    env.reset()
    # synthetic actions, generated ON THE DEVICE by a counter-based stream keyed (seed, global env, agent, t):
    # i.i.d. steer~U(-1,1), gas~U(0,1), brake~U(0,1); one small kernel per ACT_BLOCK steps inside the timed region (the
    # stream is a pure function of t, so a block of steps can be drawn at once; two buffers: the GPU may still be
    # reading the previous block when the host enqueues the next one — same stream, so no hazard, but kept simple)
    g = torch.Generator(device=dev); g.manual_seed(1234 + rank)
    ACT_BLOCK = 16
    act = [torch.empty((ACT_BLOCK, B, N, 3), dtype=torch.float32, device=dev) for _ in range(2)]
    tstep = [0]

    def next_actions():
        t = tstep[0]; tstep[0] += 1
        blk, j = divmod(t, ACT_BLOCK)
        if j == 0:
            env.env.synth_actions(t, seed=args.action_seed, out=act[blk & 1], steps=ACT_BLOCK)
            if args.actions == "drive":             # three tiny in-place kernels per ACT_BLOCK steps
                a = act[blk & 1]
                a[..., 0].mul_(0.1); a[..., 1].fill_(1.0); a[..., 2].zero_()
        return act[blk & 1][j]
    # Steady state before anything is timed: a real rollout has its episodes ending at different steps, not all
    # B TimeLimits expiring in the same step (which would put B host track generations into one burst).  One
    # un-timed TimeLimit period in which every env is reset once, at a step drawn without replacement, leaves the
    # episode phases uniformly spread (and not correlated with the env index);
    # the timed region then sees the same number of resets (B per L steps), each with its host-side generation.
    if args.stagger:
        L = 1000
        ids = torch.randperm(B, device=dev, generator=g)      # which env gets which phase: random, as in a real rollout
        for j in range(L):
            env.step(next_actions())
            msk = ((ids * L) // B == j).to(torch.uint8)
            if bool(msk.any()):
                env.reset_envs(msk)
    for k in range(W):
        env.step(next_actions())
    env.wait_refills()
    if args.debug_bits:
        from multi_car_racing_amd import _lib as _L
        _L.check(env.env.L.mcr_debug_set(env.env.h, args.debug_bits))
    env.timing(0)
    blocked0 = env.env.blocked_s
    gen0 = env.env.episodes_generated
    env.env.rollout_stats(reset=True)
    ctr0 = env.env.debug_counters()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # The host is kept at most LOOKAHEAD steps ahead of the GPU (an RL loop would be 0 steps ahead): that way the
    # consumed-episode poll inside env.step() sees the device-side auto-resets while the rollout is still running
    # and the replacement tracks are generated + staged by the refill thread INSIDE the timed region.
    LOOKAHEAD = 16
    # Instrumentation is sampled: every HIP event costs the queue a few microseconds, so the raster launch is bracketed
    # by timing events in every TIME_EVERY-th step only (the average over those launches is `roofline.avg_launch_ms`)
    # and the look-ahead fence is one event per LOOKAHEAD // 4 steps.
    TIME_EVERY = 8
    tmask = 0 if args.no_kernel_timing else (255 if args.time_all_kernels else 4)
    FENCE = LOOKAHEAD // 4
    evs = [torch.cuda.Event(blocking=True) for _ in range(4)]       # the host sleeps at the look-ahead fence instead of spinning on a core the refill thread needs
    thr0 = thread_cpu()
    cpu0 = time.process_time()                       # CPU seconds of every thread of this process (step loop, refill thread, track generators)
    t0 = time.perf_counter()
    for k in range(K):
        if tmask:
            env.timing(tmask if k % TIME_EVERY == 0 else 0)
        env.step(next_actions())
        if args.sync_every > 0 and k % args.sync_every == args.sync_every - 1:
            torch.cuda.current_stream().synchronize()
        if k % FENCE == FENCE - 1:
            j = (k // FENCE) % 4
            if k >= LOOKAHEAD:
                while not evs[j].query():           # (hipEventSynchronize spins on this runtime even for a blocking event: 0.8 of a core
                    time.sleep(1e-4)                #  that the track generator needs when a rank has two of them — query + sleep: 0.02)
            evs[j].record()
    torch.cuda.synchronize()
    t_gpu_done = time.perf_counter()
    env.wait_refills()
    closing_wait = time.perf_counter() - t_gpu_done      # the host's last tracks: the window is charged with every track its own resets consumed
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    closing_detail = None
    if args.refill == "native":
        dbg = np.zeros(8, np.int64)
        if env.env.L.mcr_refill_debug(env.env.h, dbg.ctypes.data_as(ctypes.c_void_p)) == 0:
            closing_detail = dict(zip(("queued", "in_flight", "done_unstaged", "cycles", "generated_by_waiter", "us_in_cycles", "us_total", "generator_threads"), (int(v) for v in dbg)))
    host_cores = (time.process_time() - cpu0) / elapsed
    thr1 = thread_cpu()
    me = str(threading.get_native_id())
    busy = sorted(((v - thr0.get(key, 0.0)) / elapsed, key) for key, v in thr1.items())[::-1][:6]
    # (threads that have exited by now — the track generator's, started per batch — are in host_cores_busy but not listed here)
    host_threads = [{"thread": ("step loop" if key[0] == me else key[1]) + ":" + key[0], "cores": round(v, 2)} for v, key in busy if v >= 0.01]
    ms, nl = env.timing_read()
    env.timing(0)
    env.wait_refills()
    generated = env.env.episodes_generated - gen0
    episodes, return_sum = env.env.rollout_stats()            # counted on the device: episodes that ended in the timed region
    m = reduce_metrics(B * K, elapsed, episodes=episodes, return_sum=return_sum)

    # roofline of the dominant kernel (view raster): algorithmic bytes per env-step:
    #   N*27648 (obs write) + 36*P (road quads read once per env) + 76*N (car transforms+phases) + 48*N (camera+HUD)
    P_mean = float(env.env.episode_info[:, 1].mean())
    bytes_per_env_step = N * 27648 + 36 * P_mean + 124 * N
    roofline = None
    if args.obs and nl[2] > 0:
        avg_ms = ms[2] / nl[2]
        # the timed launch is the main raster launch: envs routed to the internal streams (car<->car contact, deferred
        # position loops, re-spawns; ~21 of 4096 per step) are drawn by small launches of their own and do not count here
        ctr1 = env.env.debug_counters()
        off_main = (float((ctr1[0] - ctr0[0]) + (ctr1[2] - ctr0[2])) + float(episodes)) / K   # deferred + contact + re-spawned envs of this rank
        main_envs = max(B - off_main, 1.0)
        achieved = bytes_per_env_step * main_envs / (avg_ms * 1e-3) / 1e9
        # HBM bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE need runs of their own:
        # tools/profile_round.sh); this run did not collect counters, so the figure is quoted from the committed
        # summary of the same command and labelled with its source
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "view_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = "profiles/view_traffic.json (%s; PMC passes of tools/profile_round.sh, not measured in this run)" % tj.get("round", "?")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "k_view", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                    "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_source, "avg_launch_ms": avg_ms, "launches": int(nl[2]), "timed": "every %dth step of the timed region; HIP events that take the dispatch's own begin / end timestamps (hipExtLaunchKernelGGL start / stop events on the launch stream)" % TIME_EVERY,
                    "algorithmic_bytes_per_launch": bytes_per_env_step * main_envs, "envs_per_launch": main_envs}
    if rank == 0:
        out = {
            "metric": "env-steps/sec (num_agents=%d, 96x96 RGB obs) at batch=%d; %d GPU" % (N, B, world),
            "value": m["env_steps"] / m["elapsed_s"], "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": m["elapsed_s"] / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 rigid-body state / f64 tyre model / u8 pixels", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: num_agents=%d, batch=%d envs/GPU, %s, %s rollout, "
                                   "TimeLimit 1000 auto-reset incl. host track generation%s" % (N, B, "96x96 RGB obs" if args.obs else "obs=none", "random-action" if args.actions == "random" else "DRIVING-action (gas 1, brake 0, steer noise +-0.1)", ", episode phases staggered (steady state)" if args.stagger else ", all episodes in phase"),
                       "global_batch": B * world, "parallelism": "env-sharded dp%d (no data-path collective)" % world,
                       "episodes_reset_in_timed_region": m["episodes"], "mean_episode_return_per_env": (m["return_sum"] / m["episodes"]) if m["episodes"] else None,
                       "tracks_generated_on_host_in_timed_region_rank0": generated,
                       "env_steps_frozen_waiting_for_host_rank0": int(env.env.debug_counters()[3] - ctr0[3]),
                       "touch_verdict_mismatches_rank0": env.env.verdict_mismatches(),
                       "status_words_rank0": {k: int(v) for k, v in zip(("waits_given_up", "verdict_mismatches", "manifold_overflows", "event_overflows", "envs_frozen"), env.env.status_words()[:5])},
                       "contact_pass_beside_dynamics": bool(env.env.L.mcr_concurrent_collide(env.env.h)),
                       "stepping_loop": ("synchronised every %d step(s)" % args.sync_every) if args.sync_every > 0 else "free-running, at most 16 steps ahead of the GPU",
                       "world": "fresh world per episode (rounds 1-5)" if args.fresh_world else "one b2World per env across its episodes (the reference; csrc/k_world.h)",
                       "contact_envs_per_step_rank0": float(env.env.debug_counters()[2] - ctr0[2]) / K,
                       "deferred_envs_per_step_rank0": float(env.env.debug_counters()[0] - ctr0[0]) / K},
            "roofline": roofline,
        }
        out["config"]["step_blocked_on_refill_s_rank0"] = env.env.blocked_s - blocked0
        out["config"]["closing_wait_for_host_tracks_ms_rank0"] = round(closing_wait * 1e3, 3)
        if closing_detail is not None:
            out["config"]["closing_wait_detail_rank0"] = closing_detail
        out["config"]["host_threads_busy_rank0"] = host_threads               # by thread name (python = the step loop and the refill thread)
        out["config"]["host_cores_busy_rank0"] = round(host_cores, 2)      # CPU time / wall time of the timed region: what one rank asks of the host
        out["config"]["stream_ordering"] = {1: "phase words", 3: "phase words", 2: "events (stop events)", 0: "events", 4: "events (queues shared with another handle)", 6: "events (stop events; queues shared with another handle)"}.get(int(env.env.L.mcr_step_ordering_for(env.env.h, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))), "?") if args.streams != 1 else "single stream"
        if emu:
            out["config"]["emulated_host_share"] = emu
        if args.rccl and world == 1:
            out["config"]["process_group"] = "1-rank RCCL group (backend %s): metrics all-reduced over it" % dist.get_backend()
        if K < 200:
            out["config"]["note"] = ("short run: %d steps = %.0f ms of timed work; the default (1000 steps = one TimeLimit period, "
                                     "every env resets once) is the representative figure" % (K, m["elapsed_s"] * 1e3))
        if args.time_all_kernels:
            out["kernel_ms"] = {"collide": ms[0] / max(nl[0], 1), "dynamics": ms[1] / max(nl[1], 1), "view": ms[2] / max(nl[2], 1),
                                "collide_reset_pass": ms[3] / max(nl[3], 1), "dynamics_reset_pass": ms[4] / max(nl[4], 1),
                                "side_dynamics": ms[5] / max(nl[5], 1), "side_view": ms[6] / max(nl[6], 1), "side_reset_pass_kernel": ms[7] / max(nl[7], 1)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, bool(args.obs))
        line = json.dumps(out)
    env.close()
    if world > 1 or args.rccl:
        dist.destroy_process_group()
    if rank == 0:
        # the ONE json line is the last thing on stdout: RCCL's banner (printf into libc's buffer, flushed at exit when stdout is a pipe or a
        # file) would otherwise land behind it
        sys.stdout.flush(); ctypes.CDLL(None).fflush(None)
        print(line); sys.stdout.flush()


if __name__ == "__main__":
    main()
