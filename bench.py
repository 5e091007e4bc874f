#!/usr/bin/env python3
"""bench.py — control-steps/sec (MPC + WBC) of the MI355X hot path on synthetic Solo12 states.

One "step" = one MPC::run + one wbc_controller.compute for every instance of the batch (ratio 1:1,
SURVEY.md §8(d)).  Default workload = BASELINE.json configs[2]: batch 4096, N = 16, trot, MPC + QPWBC +
InvKin on one MI355X; `--n-steps 32 --gaits walk,trot,bounding` is configs[3].

`--gpus N` (BASELINE configs[4]): one process per GPU, every rank owns its own `--batch` instances (weak scaling)
and, per step, all-gathers the joint torques of every robot (12 f64 per instance) over RCCL on a side stream,
overlapped with the next step.  The driver starts the ranks with torch.distributed.run; run by hand without
WORLD_SIZE in the environment, `--gpus N` starts the N ranks itself (fresh child processes, nothing re-exec'd).

Prints ONE JSON line (rank 0): metric/value/unit/... plus
  roofline     — dominant kernel (mpc_solve_kernel): algorithmic FP64 flops per launch (ADMM iteration counts of
                 EVERY timed launch x the per-iteration figure of SURVEY §8(d) as a function of N) / the launch
                 durations measured with HIP events on the launch stream, against the gfx950 FP64 peak;
  accuracy     — the metric's second half: GPU torques / forces against the CPU oracle on the same seeded sample;
  cpu_baseline — the CPU oracle (oracle/, "port", -O3 -march=native, OpenMP) timed on a bounded sample of the
                 same workload on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in the order of first use; this
# process uses more than four over its legs, and two stream groups that land on one queue serialise (DESIGN.md 4.1:
# 600 k instead of 970 k steps/s).  Read by the runtime at its first call: set before anything touches the GPU.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# the host cores this process may use, read before any import that starts an OpenMP runtime (a runtime told to bind its threads
# narrows the initial thread's mask, and with it what sched_getaffinity reports later)
ALLOWED_CPUS = len(os.sched_getaffinity(0))

import numpy as np  # noqa: E402

PEAK_FP64 = 78.6e12  # gfx950 FP64 vector = matrix peak
PEAK_HBM = 8.0e12


# ---- algorithmic work per unit (scripts/alg_work.py derives the N-dependence) ----
def f_iter(N):
    """flops per ADMM iteration of one MPC instance: 4 nnzL(N) + 4 nnz(A) + 12 (n + m) with nnzL(N) = 726 N - 670
    (symbolic Cholesky of the time-interleaved reduced KKT; 10 946 at N = 16), nnz(A) = 126 N - 18, n + m = 68 N."""
    return 4.0 * (726 * N - 670) + 4.0 * (126 * N - 18) + 12.0 * 68 * N


def f_fac(N):
    """flops per KKT factorisation: sum of squared column counts, linear in N, 0.33 Mflop at N = 16 (SURVEY's figure)."""
    return 0.33e6 * (21034.0 * N - 25202.0) / (21034.0 * 16 - 25202.0)


F_ASM = 3.0e3        # assembly (M7 + M8 structural)
F_WBC = 15.0e3       # kinematics + 2x RNEA + InvKin + QP build
F_WBC_IT = 1.0e3     # per 12-variable ADMM iteration


def roofline_block(B, N, N_gait, launch_ms, iters, gaits=None, kernel=None, wbc_too=True):
    """The `roofline` object of one bench leg: the MPC kernel's algorithmic flops per launch -- ADMM iteration
    counts of EVERY timed launch x f_iter(N), plus one factorisation and assembly per instance -- over the launch durations
    measured with HIP events on the launch stream, against the FP64 peak; HBM traffic per launch from the committed rocprofv3
    --pmc passes of the same shape (null where that shape has not been profiled on these kernel sources).
    launch_ms: (K,) durations, iters: (K, B) iteration counts."""
    launch_ms = np.asarray(launch_ms, dtype=np.float64)
    iters = np.asarray(iters, dtype=np.float64).reshape(len(launch_ms), -1)
    K = len(launch_ms)
    flops = iters.sum(axis=1) * f_iter(N) + B * (f_fac(N) + F_ASM)   # per launch
    achieved = float(flops.sum() / (launch_ms.sum() * 1e-3))
    bytes_launch = B * b_alg(N, N_gait)
    traffic, traffic_src = pmc_traffic_bytes(B, N, gaits)
    if kernel is None:
        kernel = "mpc_solve_kernel<%d,%s,false,%s>" % (1 if N <= 16 else 2, "true" if N in (16, 32) else "false",
                                                       "true" if (N > 16 and B > 512) else "false")
    return {
        "kernel": kernel, "bound": "fp64-valu-issue", "achieved": achieved / 1e12, "peak": PEAK_FP64 / 1e12,
        "unit": "TFLOP/s", "frac": achieved / PEAK_FP64, "traffic": traffic, "traffic_source": traffic_src,
        "launch_ms_mean": float(launch_ms.mean()), "launch_ms_min": float(launch_ms.min()), "launch_ms_max": float(launch_ms.max()),
        "launches": int(K), "mean_admm_iters": float(iters.mean()), "max_admm_iters": int(iters.max()),
        "flops_per_iteration": f_iter(N), "flops_per_factorisation": f_fac(N),
        "algorithmic_bytes_per_launch": bytes_launch,
        "hbm_frac_algorithmic": (bytes_launch / (launch_ms.mean() * 1e-3)) / PEAK_HBM,
        "note": "roof = FP64 peak of gfx950 (vector = matrix = 78.6 TFLOP/s); neither of the contract's two labels fits: the "
                "kernel is bound by FP64 VALU instruction issue on dependent chains (one wavefront per SIMD, no MFMA issued, "
                "HBM at a fraction of a per cent), DESIGN.md 4.1"}


def b_alg(N, N_gait):
    """compulsory HBM bytes per control step: inputs + outputs + persisted solver state read and written once."""
    io = 8 * (12 * (N + 1) + 12 * N_gait + 89 + 24 * N + 48)
    state = 2 * 8 * (24 * N + 44 * N + 44 * N + 4)
    return float(io + state)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--n-steps", type=int, default=16, help="MPC horizon")
    ap.add_argument("--gaits", type=str, default="trot")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-collective", action="store_true", help="N > 1: skip the per-step all-gather of torques")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figures (profiling runs)")
    ap.add_argument("--no-configs", action="store_true", help="skip the in-line legs of the other BASELINE configurations "
                    "(batch 1, batch 256 MPC-only, batch 4096 N=32 mixed gaits)")
    ap.add_argument("--dry-ranks", type=int, default=0, metavar="N",
                    help="multi-GPU pre-flight on ONE GPU: walk every per-rank code path of --gpus N (handle creation + "
                         "self-test, this rank's shard of the inputs, a short timed region, JSON assembly) for LOCAL_RANK = "
                         "0..N-1, one rank at a time, all mapped to device 0; prints one line with per-rank figures")
    ap.add_argument("--cpu-sample", type=int, default=1024, help="instances in the CPU baseline / accuracy sample")
    ap.add_argument("--cpu-steps", type=int, default=24, help="control steps of the CPU baseline / accuracy sample")
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads for the CPU baseline (a 1-GPU box owns 16 cores)")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` by hand: start N fresh rank processes (no exec of this one, no GPU touched here)."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies early would leave the others waiting in the rendezvous: poll, and end the survivors (exact PIDs)
    import threading

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    sys.stdout.write(b"".join(out0).decode())
    sys.stdout.flush()
    return 0 if (not failed and all(rc == 0 for rc in rcs)) else 1


class StubEngine:
    """QRW_BENCH_STUB=1: NO kernels — a rehearsal of the multi-rank plumbing (sharding, per-step all-gather pipeline,
    barrier + max-over-ranks timing, JSON assembly) on CPU tensors for the world-size-2 gloo test.  The line it prints
    is marked "data": "stub" and its value means nothing."""

    def __init__(self, B, N, b0, torch):
        self.B, self.N, self.b0, self.torch = B, N, b0, torch
        self._s = 0

    def mpc_solve(self, xref, fsteps, s, out=None):
        out.zero_()
        self._s = s
        return out

    def wbc_compute(self, q, dq, f_cmd, contacts, pg, vg, ag, out=None):
        t = self.torch
        if out is None:
            out = dict(tau_ff=t.empty((self.B, 12), dtype=t.float64), f_with_delta=t.zeros((self.B, 12), dtype=t.float64))
        idx = t.arange(self.b0, self.b0 + self.B, dtype=t.float64)
        out["tau_ff"].copy_(idx[:, None] * 1e-3 + self._s + t.arange(12, dtype=t.float64)[None, :] * 1e-6)
        return out

    def mpc_stats(self):
        return dict(iters=np.full(self.B, 25, np.int32), status=np.ones(self.B, np.int32))

    def wbc_stats(self):
        return dict(iters=np.full(self.B, 25, np.int32))


def main():
    args = parse_args()
    if args.dry_ranks > 0:
        raise SystemExit(dry_ranks(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (the launcher and the flag must agree)" % (args.gpus, world))

    import torch

    stub = os.environ.get("QRW_BENCH_STUB") == "1"
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # rehearsal knobs (1-GPU box): QRW_SINGLE_DEVICE=1 puts every rank on cuda:0, QRW_DIST_BACKEND=gloo avoids RCCL
    if os.environ.get("QRW_SINGLE_DEVICE") == "1":
        local_rank = 0
    if stub:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    dist = None
    backend = None
    # QRW_FORCE_COLLECTIVE=1: run the N > 1 code path (process group, per-step all-gather pipeline, both timed regions)
    # with however many ranks there are -- one RCCL rank on a 1-GPU box exercises the real nccl calls of that path
    multi = world > 1 or os.environ.get("QRW_FORCE_COLLECTIVE") == "1"
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("QRW_DIST_BACKEND", "gloo" if stub else "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import synth
    from sharding import TorqueGatherPipeline

    B, N = args.batch, args.n_steps
    N_gait = max(20, N + 4)
    W, K = max(args.warmup, 1), args.steps  # the first call (num_iter == 0) is the QP setup: always untimed
    gaits = tuple(args.gaits.split(","))
    collective = multi and not args.no_collective
    n_regions = 2 if collective else 1      # N > 1: a second timed region without the all-gather, same line
    n_in = W + n_regions * K
    # --dry-ranks: this process stands in for rank QRW_DRY_RANK of a QRW_DRY_WORLD-rank job (its shard of the instances)
    shard_rank = int(os.environ.get("QRW_DRY_RANK", rank))
    sb = synth.SyntheticBatch(B, N, N_gait=N_gait, gaits=gaits, n_seq=n_in, b0=shard_rank * B)
    t_gen = time.time()
    steps = [sb.step(s) for s in range(n_in)]
    t_gen = time.time() - t_gen

    def dev_t(key):
        return [torch.from_numpy(np.ascontiguousarray(st[key])).to(dev) for st in steps]

    xref, fsteps = dev_t("xref"), dev_t("fsteps")
    q, dq, contacts = dev_t("q"), dev_t("dq"), dev_t("contacts")
    pg, vg, ag = dev_t("pgoals"), dev_t("vgoals"), dev_t("agoals")

    t_create = time.time()
    if stub:
        eng = StubEngine(B, N, rank * B, torch)
    else:
        import qrw_hip
        eng = qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, dt_mpc=0.02, T_gait=0.02 * N, dt_wbc=0.002, device=local_rank)
    t_create = time.time() - t_create  # includes this process's known-answer self-test of the kernel (first handle on the device)
    mpc_out = torch.empty((B, 24, N), dtype=torch.float64, device=dev)
    f_cmd = torch.empty((B, 12), dtype=torch.float64, device=dev)
    wbc_bufs = [None, None]  # two output sets: the gather of step s reads one while step s+1 writes the other
    pipe = TorqueGatherPipeline(B, dev) if multi else None
    mk_ev = (lambda: torch.cuda.Event(enable_timing=True)) if not stub else (lambda: None)
    ev = [(mk_ev(), mk_ev()) for _ in range(n_regions * K)]
    ev_w = [(mk_ev(), mk_ev()) for _ in range(n_regions * K)]
    it_dev = torch.zeros((n_regions * K, B), dtype=torch.int32, device=dev) if not stub else None

    def one_step(s, timed_idx=None, gather=False):
        rec = (timed_idx is not None) and not stub
        if rec:
            ev[timed_idx][0].record()
        eng.mpc_solve(xref[s], fsteps[s], s, out=mpc_out)
        if rec:
            ev[timed_idx][1].record()
        f_cmd.copy_(mpc_out[:, 12:, 0])
        i = s & 1
        if gather:
            pipe.wait_buffer_free(i)  # the gather that last read this output set (two steps ago) has finished
        if rec:
            ev_w[timed_idx][0].record()
        wbc_bufs[i] = eng.wbc_compute(q[s], dq[s], f_cmd, contacts[s], pg[s], vg[s], ag[s], out=wbc_bufs[i])
        if rec:
            ev_w[timed_idx][1].record()
            eng.copy_mpc_iters(it_dev[timed_idx])  # device-to-device, stays in HBM; read back after the timed region
        if gather:
            pipe.issue(i, wbc_bufs[i]["tau_ff"])   # all-gather on the side stream, overlaps the next step

    def barrier():
        if multi:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize()

    def timed_region(first, base_idx, gather):
        barrier()
        t0 = time.perf_counter()
        for i in range(K):
            one_step(first + i, base_idx + i, gather)
        if gather:
            pipe.drain()  # the last all-gather belongs to the timed work
        barrier()
        el = time.perf_counter() - t0
        local_el.append(el)  # this rank's own clock, before the maximum over the ranks (a straggler GPU shows in per_rank)
        if multi:
            tt = torch.tensor([el], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    local_el = []

    ranks_seen = [0]
    if multi:
        me = torch.tensor([rank], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        seen = [torch.zeros_like(me) for _ in range(world)]
        dist.all_gather(seen, me)
        ranks_seen = sorted(int(t.item()) for t in seen)
    for s in range(W):
        one_step(s, gather=collective)
    if collective:
        pipe.drain()
    elapsed = timed_region(W, 0, collective)
    gather_ok = None
    if collective:
        gather_ok = pipe.check_last(rank, world, wbc_bufs[(W + K - 1) & 1]["tau_ff"])
        elapsed_nc = timed_region(W + K, K, False)

    total_steps = world * B * K
    value = total_steps / elapsed
    out = {
        "metric": "control-steps/sec (MPC+WBC)", "value": value, "unit": "steps/s", "n_gpus": world,
        "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "stub" if stub else "synthetic",
        "config": {"workload": "Solo12 %s, batch %d per GPU, horizon N=%d, MPC (OSQP-style ADMM) + WBC (InvKin + "
                               "RNEA + box-QP) per control step, ratio 1:1; open-loop sequence: every call's current state "
                               "is the reference plus fresh noise (harder than the closed receding-horizon sequence of "
                               "SURVEY 8(d), which is reported beside it as closed_loop_sequence)" % ("/".join(gaits), B, N),
                   "batch_per_gpu": B, "n_steps": N, "gaits": list(gaits), "parallelism": "batch-sharded x%d" % world},
    }
    if multi:
        out["collective"] = {
            "op": "all_gather of joint torques, one per control step, side stream, overlapped with the next step"
                  if collective else "none (--no-collective)",
            "backend": ("RCCL (torch.distributed nccl)" if backend == "nccl" else backend),
            "bytes_per_rank_per_step": B * 12 * 8, "ranks_seen": ranks_seen, "gathered_block_check": gather_ok,
            "no_collective_steps_per_s": (total_steps / elapsed_nc) if collective else None,
            "no_collective_ms_per_step": (1e3 * elapsed_nc / K) if collective else None}

    if not stub:
        # per-kernel figures (this rank): durations from HIP events on the launch stream, iteration counts of EVERY timed launch
        mpc_ms = np.array([a.elapsed_time(b) for a, b in ev[:K]])
        wbc_ms = np.array([a.elapsed_time(b) for a, b in ev_w[:K]])
        iters = it_dev[:K].cpu().numpy().astype(np.float64)          # (K, B)
        ms, ws = eng.mpc_stats(), eng.wbc_stats()
        out["roofline"] = roofline_block(B, N, N_gait, mpc_ms, iters, gaits, kernel="mpc_solve_kernel")
        out["kernels_ms"] = {"mpc_solve_kernel": float(mpc_ms.mean()), "wbc_kernel": float(wbc_ms.mean())}
        out["solver"] = {"mpc_solved_last_step": int((ms["status"] == 1).sum()), "mpc_instances": B,
                         "wbc_mean_iters": float(ws["iters"].mean())}
        out["input_gen_s"] = t_gen
        out["handle_create_s"] = t_create
        out["device_ordinal"] = local_rank
        out["shard_first_instance"] = shard_rank * B
    if multi:
        # what a first scaling run is judged on, from EVERY rank (rank 0's own figures alone would hide a straggler GPU)
        mine = [float(rank), float(local_rank), float(shard_rank * B), local_el[0], local_el[1] if len(local_el) > 1 else float("nan"),
                float(mpc_ms.mean()) if not stub else float("nan"), float(wbc_ms.mean()) if not stub else float("nan"),
                float(iters.mean()) if not stub else float("nan"), float(len(ranks_seen)),
                (1.0 if gather_ok else 0.0) if gather_ok is not None else float("nan")]
        tv = torch.tensor(mine, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        allv = [torch.zeros_like(tv) for _ in range(world)]
        dist.all_gather(allv, tv)
        nn = lambda x: None if x != x else x
        out["per_rank"] = [
            {"rank": int(v[0]), "device_ordinal": int(v[1]), "shard_first_instance": int(v[2]), "timed_region_s": float(v[3]),
             "steps_per_s": B * K / float(v[3]), "no_collective_steps_per_s": nn(B * K / float(v[4]) if float(v[4]) == float(v[4]) else float("nan")),
             "launch_ms_mean": nn(float(v[5])), "wbc_ms_mean": nn(float(v[6])), "mean_admm_iters": nn(float(v[7])),
             "ranks_seen": int(v[8]), "gathered_block_check": (None if float(v[9]) != float(v[9]) else bool(v[9]))}
            for v in (t.cpu().tolist() for t in allv)]
    if rank == 0 and world == 1 and not stub:
        out["mpc_solves_per_s"] = B * K / (mpc_ms.sum() * 1e-3)
        out["wbc_steps_per_s"] = B * K / (wbc_ms.sum() * 1e-3)
        if not args.no_secondary:
            out["two_stream_groups"] = stream_groups_figure(B, N, N_gait, dev, W, K, dict(
                xref=xref, fsteps=fsteps, q=q, dq=dq, contacts=contacts, pgoals=pg, vgoals=vg, agoals=ag))
            out["pipelined_sequence"] = pipelined_sequence_figure(B, N, N_gait, dev, W, K, dict(
                xref=xref, fsteps=fsteps, q=q, dq=dq, contacts=contacts, pgoals=pg, vgoals=vg, agoals=ag))
            out["closed_loop_sequence"] = closed_loop_sequence(B, N, N_gait, gaits, dev, W, K)
            out["secondary_ratio_1_10"] = device_resident_loop(sb, B, N, N_gait, dev)
            if B % 2 == 0:  # the same loop with the fleet as two stream groups (Controller_batch(groups=2), opt-in)
                out["secondary_ratio_1_10_two_groups"] = device_resident_loop(sb, B, N, N_gait, dev, groups=2)
                out["secondary_ratio_1_10_two_groups_free"] = device_resident_loop(sb, B, N, N_gait, dev, groups=2, free_running=True)
                out["secondary_ratio_1_10_two_groups_staggered"] = device_resident_loop(sb, B, N, N_gait, dev, groups=2, free_running=True,
                                                                                        stagger=True)
            # Controller_batch's default split: one SIMD per wavefront of the loop's kernels (64 compute units at batch 4096), the
            # latency-oriented setting the mode exists for
            out["secondary_ratio_1_10_async"] = device_resident_loop(sb, B, N, N_gait, dev, multiprocessing=True)
            # 32 compute units for the loop's stream: the highest free-running rate, paid for in iteration latency.  The caller works
            # on the loop's own stream here (Controller_batch.loop_stream, the recommended use): with a THIRD stream for the caller's
            # work the figure depends on which hardware queue that stream happens to share -- by rounds 5-6 this process has created
            # so many streams before this leg that the caller's landed on the MPC stream's queue (7.9 M iterations/s, 3.3 ms worst:
            # every caller-side copy queued behind the running solve), while the same leg alone in a fresh process measures 12.8 M /
            # 0.34 ms (scripts/gpu_async_cus.py, docs/HISTORY.md 8).  INTEGRATION.md: GPU_MAX_HW_QUEUES.
            out["secondary_ratio_1_10_async_32cu"] = device_resident_loop(sb, B, N, N_gait, dev, multiprocessing=True, loop_cus=32,
                                                                          on_loop_stream=True)
            # two staggered stream groups through compute() (joined on the caller's stream every iteration): what
            # Controller_batch.for_deadline builds for fleets of 1025..2048 robots; opt-in (the default object is one handle:
            # secondary_ratio_1_10 above)
            if B % 2 == 0:
                out["secondary_ratio_1_10_two_groups_staggered_joined"] = device_resident_loop(sb, B, N, N_gait, dev, groups=2, stagger=True)
            if (N, gaits) == (16, ("trot",)):
                out["realtime_slot"] = realtime_slot_leg(sb, N, N_gait, dev)
        if not args.no_configs and not args.no_secondary and (B, N, gaits) == (4096, 16, ("trot",)):
            # the metric reads "batch {1, 256, 4096}" and BASELINE lists configs 2 and 4: every single-GPU figure in this ONE line
            # each leg carries the metric's accuracy half too: the HIP path against the CPU oracle on a bounded sample of the
            # leg's own workload (first `acc[0]` instances x `acc[1]` control steps), outside every timed region
            thr = args.cpu_threads
            out["batch_1"] = config_leg(1, 16, ("trot",), dev, W=3, K=20, acc=(1, 23), threads=thr)
            out["batch_256_mpc_only"] = config_leg(256, 16, ("trot",), dev, W=3, K=20, mpc_only=True, acc=(256, 8), threads=thr)
            out["config4_n32_mixed"] = config_leg(4096, 32, ("walk", "trot", "bounding"), dev, W=3, K=6, closed=True, groups=2,
                                                  acc=(128, 6), threads=thr)
        if not args.no_cpu_baseline:
            # bounded sample: --cpu-sample instances are sized for the 16 host cores a 1-GPU box usually owns (~25 s of CPU work);
            # a box that gives this process fewer cores gets proportionally fewer instances, so the leg's wall time stays put
            cores = max(1, min(int(args.cpu_threads), ALLOWED_CPUS))
            Bc = max(32, min(args.cpu_sample, (args.cpu_sample * cores + 15) // 16))
            base, ref_out = cpu_baseline(synth, Bc, N, N_gait, gaits, args.cpu_threads, args.cpu_steps)
            out["cpu_baseline"] = base
            out["accuracy"] = accuracy_vs_oracle(synth, ref_out, Bc, N, N_gait, gaits, args.cpu_steps, dev)
            out["torque_max_abs_err"] = out["accuracy"]["torque_max_abs_err"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()


def config_leg(B, N, gaits, dev, W, K, mpc_only=False, closed=False, groups=0, acc=None, threads=16):
    """One BASELINE configuration other than the headline's, timed in-line on a FRESH handle with the headline's method:
    W untimed warm-up steps (the first one sets the QPs up), K timed steps bracketed by synchronisation, HIP events around
    every mpc_solve launch, the ADMM iteration counts of every timed launch (device-to-device), FP64 fraction with THIS
    horizon's work terms.  closed=True adds SURVEY 8(d)'s closed receding-horizon variant of the same configuration (inputs
    generated by an untimed pass, replayed from HBM on a fresh handle), groups=S the open-loop workload as S stream groups."""
    import torch

    import qrw_hip
    import synth

    N_gait = max(20, N + 4)
    mk = dict(n_steps=N, N_gait=N_gait, dt_mpc=0.02, T_gait=0.02 * N, dt_wbc=0.002, device=dev.index or 0)
    keys = ("xref", "fsteps", "q", "dq", "contacts", "pgoals", "vgoals", "agoals")

    def to_dev(d):
        return {k: torch.from_numpy(np.ascontiguousarray(d[k])).to(dev) for k in keys}

    def timed(seq, label):
        eng = qrw_hip.Batch(B, **mk)
        mpc_out = torch.empty((B, 24, N), dtype=torch.float64, device=dev)
        f_cmd = torch.empty((B, 12), dtype=torch.float64, device=dev)
        it_dev = torch.zeros((K, B), dtype=torch.int32, device=dev)
        st_dev = torch.zeros((K, B), dtype=torch.int32, device=dev)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        w = None

        def step(s, i=None):
            nonlocal w
            t = seq[s]
            if i is not None:
                ev[i][0].record()
            eng.mpc_solve(t["xref"], t["fsteps"], s, out=mpc_out)
            if i is not None:
                ev[i][1].record()
            if not mpc_only:
                f_cmd.copy_(mpc_out[:, 12:, 0])
                w = eng.wbc_compute(t["q"], t["dq"], f_cmd, t["contacts"], t["pgoals"], t["vgoals"], t["agoals"], out=w)
            if i is not None:
                eng.copy_mpc_iters(it_dev[i])

        for s in range(W):
            step(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            step(W + i, i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ms = np.array([a.elapsed_time(b) for a, b in ev])
        iters = it_dev.cpu().numpy().astype(np.float64)
        status = eng.mpc_stats()["status"]
        sl = eng.mpc_slice_stats()  # (last timed launch; zeros where the launch is not time-sliced)
        eng.close()
        flops = iters.sum() * f_iter(N) + K * B * (f_fac(N) + F_ASM)
        extra = {}
        if sl["levels"]:
            extra["time_slicing"] = {"slice_iterations": sl["chunk"], "priority_levels": sl["levels"],
                                     "parks_per_level_last_launch": [int(v) for v in sl["parks_per_level"][:sl["levels"]]],
                                     "what": "solves parked after a slice and taken longest-predicted-remainder first (DESIGN.md 4.1); "
                                             "level 0 = first parks"}
        return {**extra, "value": B * K / el, "unit": "MPC solves/s" if mpc_only else "steps/s", "ms_per_step": 1e3 * el / K,
                "launch_ms_mean": float(ms.mean()), "mean_admm_iters": float(iters.mean()), "max_admm_iters": int(iters.max()),
                "max_iter_exit_share": float((iters >= 4000).mean()),
                "solved_share_last_step": float((status == 1).mean()),
                "roofline_frac": float(flops / (ms.sum() * 1e-3) / PEAK_FP64), "flops_per_iteration": f_iter(N),
                "roofline": roofline_block(B, N, N_gait, ms, iters, gaits if label.startswith("open-loop") else None),
                "timed_steps": K, "warmup_steps": W, "sequence": label}

    sb = synth.SyntheticBatch(B, N, N_gait=N_gait, gaits=gaits, n_seq=W + K)
    seq = [to_dev(sb.step(s)) for s in range(W + K)]
    res = timed(seq, "open-loop noisy states (as the headline)")
    res["workload"] = "Solo12 %s, batch %d, horizon N=%d, %s" % ("/".join(gaits), B, N, "MPC only" if mpc_only else "MPC + WBC 1:1")
    if groups and B % groups == 0:
        grp = qrw_hip.StreamGroups(B, groups=groups, **mk)

        def gstep(s):
            t = seq[s]
            grp.control_step(t["xref"], t["fsteps"], s, t["q"], t["dq"], t["contacts"], t["pgoals"], t["vgoals"], t["agoals"])

        for s in range(W):
            gstep(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            gstep(W + i)
        torch.cuda.synchronize()
        res["stream_groups_%d_steps_per_s" % groups] = B * K / (time.perf_counter() - t0)
        grp.close()
    if closed:
        gen = qrw_hip.Batch(B, **mk)
        sbc = synth.SyntheticBatch(B, N, N_gait=N_gait, gaits=gaits, n_seq=W + K)
        cseq, x0 = [], None
        for s in range(W + K):
            t = to_dev(sbc.step(s, x0))
            x0 = gen.mpc_solve(t["xref"], t["fsteps"], s)[:, :12, 0].cpu().numpy()
            cseq.append(t)
        gen.close()
        c = timed(cseq, "closed receding-horizon sequence (SURVEY 8(d))")
        res["closed_loop"] = {k: c[k] for k in ("value", "unit", "ms_per_step", "launch_ms_mean", "mean_admm_iters", "max_admm_iters",
                                                "max_iter_exit_share", "roofline_frac", "roofline")}
    if acc:
        res["accuracy"] = leg_accuracy(acc[0], acc[1], N, N_gait, gaits, mpc_only, dev, threads)
    return res


def leg_accuracy(Bc, steps, N, N_gait, gaits, mpc_only, dev, threads):
    """The metric's accuracy half for one configuration leg: the first Bc instances of the leg's workload (instance seeds are
    per robot, so these ARE the leg's robots) x `steps` control steps from the set-up call on, HIP path against the CPU oracle
    on identical inputs: MPC result, and -- unless the leg is MPC-only -- joint torques and contact forces; ADMM iteration
    counts compared solve by solve.  Outside every timed region (the oracle is the checker here, never the thing measured)."""
    import torch

    import qrw_hip
    import synth

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    oracle.build(fast=True)
    cores = max(1, min(int(threads), ALLOWED_CPUS))
    if cores < 8 and Bc > 16:  # a box that gives this process few host cores: a smaller sample, the same wall time
        Bc = max(16, (Bc * cores + 7) // 8)
    sb = synth.SyntheticBatch(Bc, N, N_gait=N_gait, gaits=gaits, n_seq=steps)
    eng = qrw_hip.Batch(Bc, n_steps=N, N_gait=N_gait, dt_mpc=0.02, T_gait=0.02 * N, dt_wbc=0.002, device=dev.index or 0)
    mpc = oracle.MPCBatch(Bc, 0.02, N, 0.02 * N, N_gait, fast=True)
    wbc = None if mpc_only else oracle.WbcBatch(Bc, 0.002, fast=True)
    x_abs = x_ref = tau_abs = tau_ref = f_abs = f_ref = 0.0
    it_mismatch = solves = 0
    t_cpu = 0.0
    for s in range(steps):
        d = sb.step(s)
        t = {k: torch.from_numpy(np.ascontiguousarray(d[k])).to(dev) for k in
             ("xref", "fsteps", "q", "dq", "contacts", "pgoals", "vgoals", "agoals")}
        o = eng.mpc_solve(t["xref"], t["fsteps"], s)
        a = time.perf_counter()
        r = mpc.run(s, d["xref"], d["fsteps"], cores)
        t_cpu += time.perf_counter() - a
        og = o.cpu().numpy()
        it_ref, _ = mpc.iters()
        it_mismatch += int((eng.mpc_stats()["iters"] != it_ref).sum())
        solves += Bc
        x_abs, x_ref = max(x_abs, float(np.abs(og - r).max())), max(x_ref, float(np.abs(r).max()))
        if not mpc_only:
            w = eng.wbc_compute(t["q"], t["dq"], o[:, 12:, 0].contiguous(), t["contacts"], t["pgoals"], t["vgoals"], t["agoals"])
            tau, _, _, f = wbc.compute(d["q"], d["dq"], np.ascontiguousarray(r[:, 12:, 0]), d["contacts"], d["pgoals"],
                                       d["vgoals"], d["agoals"], cores)
            tg, fg = w["tau_ff"].cpu().numpy(), w["f_with_delta"].cpu().numpy()
            tau_abs, tau_ref = max(tau_abs, float(np.abs(tg - tau).max())), max(tau_ref, float(np.abs(tau).max()))
            f_abs, f_ref = max(f_abs, float(np.abs(fg - f).max())), max(f_ref, float(np.abs(f).max()))
    eng.close()
    rel = [x_abs / x_ref] + ([] if mpc_only else [tau_abs / tau_ref, f_abs / f_ref])
    res = {"mpc_result_max_abs_err": x_abs, "mpc_result_max_rel_err": x_abs / x_ref,
           "admm_iteration_count_mismatches": it_mismatch, "solves_compared": solves, "tolerance_rel": 1e-4,
           "within_tolerance": bool(max(rel) < 1e-4),
           "sample": "first %d instances x %d control steps (from the set-up call on) of this leg's workload, HIP path vs CPU oracle "
                     "(oracle/, %d threads, %.1f s), max over instances and steps; relative = max abs error / max abs reference"
                     % (Bc, steps, cores, t_cpu)}
    if not mpc_only:
        res.update({"torque_max_abs_err": tau_abs, "torque_max_rel_err": tau_abs / tau_ref, "force_max_rel_err": f_abs / f_ref,
                    "torque_scale_Nm": tau_ref})
    return res


def dry_ranks(args):
    """`--dry-ranks N`: the multi-GPU pre-flight of VERDICT r2 item 7.  No 8-GPU node is available to the build, so a first
    SCALE run must not fail on plumbing: for LOCAL_RANK = 0..N-1, ONE RANK AT A TIME (a 1-GPU box allows few processes on
    its card), start this script as that rank of an N-rank job with every rank mapped to device 0 and a single-rank process
    group (the real nccl init / barrier / all_gather_into_tensor / all_reduce calls of the N > 1 branch run, on one rank).
    Each child walks the rank's whole path -- qrw_create + known-answer self-test, ITS shard of the synthetic inputs
    (b0 = rank * batch), warm-up, both timed regions, JSON assembly -- and prints its line; this parent collects them."""
    import socket

    n = int(args.dry_ranks)
    per_rank = []
    t_all = time.time()
    for r in range(n):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ, QRW_DRY_RANK=str(r), QRW_DRY_WORLD=str(n), QRW_SINGLE_DEVICE="1", QRW_FORCE_COLLECTIVE="1",
                   RANK="0", LOCAL_RANK=str(r), WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        argv = [a for a in sys.argv[1:]]
        for i, a in enumerate(argv):  # drop the --dry-ranks flag (and its value)
            if a == "--dry-ranks":
                del argv[i:i + 2]
                break
            if a.startswith("--dry-ranks="):
                del argv[i]
                break
        argv += ["--gpus", "1", "--no-secondary", "--no-configs", "--no-cpu-baseline"]
        t0 = time.time()
        p = subprocess.run([sys.executable, os.path.abspath(__file__)] + argv, env=env, capture_output=True, text=True)
        wall = time.time() - t0
        line = None
        for ln in p.stdout.splitlines():
            if ln.startswith("{"):
                line = json.loads(ln)
        if p.returncode != 0 or line is None:
            sys.stderr.write(p.stderr[-2000:])
            print(json.dumps({"dry_ranks": n, "failed_rank": r, "rc": p.returncode}), flush=True)
            return 1
        per_rank.append({"local_rank": r, "device_ordinal_used": line.get("device_ordinal"), "shard_first_instance": line.get("shard_first_instance"),
                         "steps_per_s": line["value"], "launch_ms_mean": line.get("roofline", {}).get("launch_ms_mean"),
                         "input_gen_s": line.get("input_gen_s"), "handle_create_s": line.get("handle_create_s"),
                         "ranks_seen": line.get("collective", {}).get("ranks_seen"),
                         "gathered_block_check": line.get("collective", {}).get("gathered_block_check"),
                         "backend": line.get("collective", {}).get("backend"), "process_wall_s": wall})
    print(json.dumps({"dry_ranks": n, "what": "per-rank pre-flight of --gpus %d on one GPU: every rank's path walked one rank at a time on "
                      "device 0 with a one-rank RCCL group; NOT a scaling measurement" % n, "batch_per_rank": args.batch,
                      "n_steps": args.n_steps, "steps": args.steps, "per_rank": per_rank, "total_wall_s": time.time() - t_all}), flush=True)
    return 0


def mpc_source_stamp():
    """sha256 over the sources that decide mpc_solve_kernel's HBM traffic (the kernel, its sweeps, the state layout)."""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "quadruped-reactive-walking_amd", "csrc")
    for f in ("mpc_kernel.hip", "chain_sweep.h", "qrw_device.h"):
        h.update(open(os.path.join(csrc, f), "rb").read())
    k = open(os.path.join(csrc, "qrw_kernels.h")).read()  # of the shared header only the MPC state layout
    i = k.index("enum MpcStateItem")
    h.update(k[i:k.index("};", i)].encode())
    return h.hexdigest()


PMC_SHAPES = {  # (batch, horizon, gaits) of the bench legs that have committed counter passes -> tag of their summary files
    (4096, 16, ("trot",)): "bench_b4096",                                 # the headline (BASELINE config 3)
    (4096, 32, ("walk", "trot", "bounding")): "n32_mixed_time_sliced",    # config 4
    (256, 16, ("trot",)): "bench_b256",                                   # config 2
    (1, 16, ("trot",)): "bench_b1",                                       # the metric's batch-1 point
}


def pmc_traffic_bytes(B, N, gaits=None):
    """HBM bytes per mpc_solve_kernel launch.  NOT measured by this run: read from the committed rocprofv3 PMC passes of
    the same bench shape (profiles/<round>_pmc_summary_<tag>.json, separate --pmc runs as MI355X_MICROARCH.md prescribes:
    scripts/pmc_profile.sh; FETCH_SIZE / WRITE_SIZE are in KiB; the kernel's accesses are 8 bytes per lane, a width the guide
    marks uncalibrated on gfx950, so no correction factor is applied).  Only valid for a profiled shape (PMC_SHAPES: the open-loop
    workload of that batch, horizon and gait mix) AND the profiled kernel: every summary carries a stamp of the kernel's sources,
    and a summary whose stamp differs from the sources this run was built from is refused (traffic null) instead of silently
    going stale."""
    tag = PMC_SHAPES.get((int(B), int(N), tuple(gaits) if gaits is not None else None))
    if tag is None:
        return None, None
    now = mpc_source_stamp()
    refused = []
    suffix = "_pmc_summary_%s.json" % tag
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(suffix)), reverse=True):
        path = os.path.join(ROOT, "profiles", name)
        try:
            rows = json.load(open(path))
            stamp = json.load(open(path.replace("_pmc_summary_", "_pmc_stamp_"))).get("mpc_source_sha256")
        except Exception:
            refused.append("%s: no source stamp" % name)
            continue
        if stamp != now:
            refused.append("%s: collected on other kernel sources" % name)
            continue
        tot = 0.0
        for r in rows:
            if "mpc_solve_kernel" in r["kernel"] and r["counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
                tot += r["mean"] * 1024.0
        if tot:
            return tot, "profiles/%s (static: rocprofv3 --pmc passes of this shape on these kernel sources, not collected by this run)" % name
    return None, "refused: " + "; ".join(refused) if refused else None


def stream_groups_figure(B, N, N_gait, dev, W, K, data, S=2):
    """The SAME workload as the headline (same inputs, same K steps, same per-instance results) with the fleet cut into
    S independent groups, each with its own handle and stream and no cross-group synchronisation inside the timed
    region: a step's launch ends with its longest solve (2 000-2 750 ADMM iterations against a mean of ~515) while most
    of the chip is already idle; with two groups in flight one group's stragglers run beside the other group's next step.
    A deployment choice above the C ABI (two qrw handles per GPU: qrw_hip.StreamGroups), not a different kernel; three
    groups give the same, four and more less (scripts/gpu_subbatch_exp.py)."""
    import torch

    import qrw_hip

    Bs = B // S
    grp = qrw_hip.StreamGroups(B, groups=S, n_steps=N, N_gait=N_gait, dt_mpc=0.02, T_gait=0.02 * N, dt_wbc=0.002,
                               device=dev.index or 0)

    def step(s):
        grp.control_step(data["xref"][s], data["fsteps"][s], s, data["q"][s], data["dq"][s], data["contacts"][s],
                         data["pgoals"][s], data["vgoals"][s], data["agoals"][s])

    torch.cuda.synchronize()
    for s in range(W):
        step(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(W + i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    grp.close()
    return {"value": S * Bs * K / el, "unit": "steps/s", "ms_per_step": 1e3 * el / K, "groups": S, "instances_per_group": Bs,
            "what": "the headline workload with the batch split into %d independent stream groups (own handle + stream each, "
                    "no cross-group synchronisation inside the timed region): straggling solves of one group overlap the "
                    "next step of the other" % S}


def pipelined_sequence_figure(B, N, N_gait, dev, W, K, data):
    """The headline's K timed steps (same inputs, bit-identical results) submitted as ONE qrw_mpc_solve_sequence launch —
    one workgroup per task, task queues that orders each instance's consecutive solves and nothing else — followed by the
    K WBC steps.  Legitimate only where all K steps' inputs exist beforehand (log replay, open-loop sweeps: the headline's
    open-loop sequence is such a case, a closed control loop is not), hence a secondary figure: it shows what the
    device-wide barrier between the steps of independent robots costs (the launch tail of DESIGN.md 4.1)."""
    import torch

    import qrw_hip

    eng = qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, dt_mpc=0.02, T_gait=0.02 * N, dt_wbc=0.002, device=dev.index or 0)
    f_cmd = torch.empty((B, 12), dtype=torch.float64, device=dev)
    w = None
    for s in range(W):  # warm-up exactly as the headline: ordinary calls (the first one sets the QPs up)
        o = eng.mpc_solve(data["xref"][s], data["fsteps"][s], s)
        f_cmd.copy_(o[:, 12:, 0])
        w = eng.wbc_compute(data["q"][s], data["dq"][s], f_cmd, data["contacts"][s], data["pgoals"][s], data["vgoals"][s],
                            data["agoals"][s], out=w)
    # the first sequence call of a process runs the SEQ instantiation's known-answer gate (once per device and horizon, a few
    # ms): take it on a scratch handle, outside the timed region
    scratch = qrw_hip.Batch(1, n_steps=N, N_gait=N_gait, dt_mpc=0.02, T_gait=0.02 * N, dt_wbc=0.002, device=dev.index or 0)
    scratch.mpc_solve_sequence(data["xref"][0][:1].unsqueeze(0).contiguous(), data["fsteps"][0][:1].unsqueeze(0).contiguous(), 0)
    torch.cuda.synchronize()
    scratch.close()
    xs = torch.stack(data["xref"][W:W + K])
    fs = torch.stack(data["fsteps"][W:W + K])
    outs = torch.empty((K, B, 24, N), dtype=torch.float64, device=dev)
    its = torch.zeros((K, B), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.mpc_solve_sequence(xs, fs, W, out=outs, iters=its)
    for i in range(K):
        s = W + i
        f_cmd.copy_(outs[i][:, 12:, 0])
        w = eng.wbc_compute(data["q"][s], data["dq"][s], f_cmd, data["contacts"][s], data["pgoals"][s], data["vgoals"][s],
                            data["agoals"][s], out=w)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    bad = eng.mpc_sequence_timed_out()
    it = its.cpu().numpy().astype(np.float64)
    eng.close()
    flops = it.sum() * f_iter(N) + K * B * (f_fac(N) + F_ASM)
    return {"value": B * K / el, "unit": "steps/s", "ms_per_step": 1e3 * el / K, "mean_admm_iters": float(it.mean()),
            "fp64_frac_over_region": float(flops / el / PEAK_FP64), "queue_timed_out": bool(bad),
            "what": "the headline's %d timed steps as one qrw_mpc_solve_sequence launch (per-instance ordering only) + %d WBC "
                    "steps; inputs of all steps resident beforehand" % (K, K)}


def closed_loop_sequence(B, N, N_gait, gaits, dev, W, K):
    """SURVEY §8(d)'s sequence: each call's current state is the MPC's own predicted next state (scripts/test_mpc.py:78).
    The inputs of call s+1 depend on the result of call s, so the sequence is first generated with an untimed pass
    (GPU solve, host planner formulas), then replayed from HBM on a fresh handle: same inputs, same solver history, timed."""
    import torch

    import qrw_hip
    import synth

    sb = synth.SyntheticBatch(B, N, N_gait=N_gait, gaits=gaits, n_seq=W + K)
    gen = qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, dt_mpc=0.02, T_gait=0.02 * N, dt_wbc=0.002, device=dev.index or 0)
    seq, x0 = [], None
    for s in range(W + K):
        d = sb.step(s, x0)
        t = {k: torch.from_numpy(np.ascontiguousarray(d[k])).to(dev) for k in
             ("xref", "fsteps", "q", "dq", "contacts", "pgoals", "vgoals", "agoals")}
        o = gen.mpc_solve(t["xref"], t["fsteps"], s)
        x0 = o[:, :12, 0].cpu().numpy()
        seq.append(t)
    gen.close()
    eng = qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, dt_mpc=0.02, T_gait=0.02 * N, dt_wbc=0.002, device=dev.index or 0)
    mpc_out = torch.empty((B, 24, N), dtype=torch.float64, device=dev)
    f_cmd = torch.empty((B, 12), dtype=torch.float64, device=dev)
    it_dev = torch.zeros((K, B), dtype=torch.int32, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    w = None

    def step(s, i=None):
        nonlocal w
        t = seq[s]
        if i is not None:
            ev[i][0].record()
        eng.mpc_solve(t["xref"], t["fsteps"], s, out=mpc_out)
        if i is not None:
            ev[i][1].record()
        f_cmd.copy_(mpc_out[:, 12:, 0])
        w = eng.wbc_compute(t["q"], t["dq"], f_cmd, t["contacts"], t["pgoals"], t["vgoals"], t["agoals"], out=w)
        if i is not None:
            eng.copy_mpc_iters(it_dev[i])

    for s in range(W):
        step(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(W + i, i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms = np.array([a.elapsed_time(b) for a, b in ev])
    iters = it_dev.cpu().numpy().astype(np.float64)
    flops = iters.sum() * f_iter(N) + K * B * (f_fac(N) + F_ASM)
    eng.close()
    # the same replay as two stream groups (qrw_hip.StreamGroups: identical results, one group's stragglers beside the
    # other group's next step)
    grp = qrw_hip.StreamGroups(B, groups=2, n_steps=N, N_gait=N_gait, dt_mpc=0.02, T_gait=0.02 * N, dt_wbc=0.002,
                               device=dev.index or 0)

    def gstep(s):
        t = seq[s]
        grp.control_step(t["xref"], t["fsteps"], s, t["q"], t["dq"], t["contacts"], t["pgoals"], t["vgoals"], t["agoals"])

    for s in range(W):
        gstep(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        gstep(W + i)
    torch.cuda.synchronize()
    el2 = time.perf_counter() - t0
    same = bool(torch.equal(grp.mpc_out, mpc_out))
    grp.close()
    return {"value": B * K / el, "unit": "steps/s", "ms_per_step": 1e3 * el / K, "mean_admm_iters": float(iters.mean()),
            "max_admm_iters": int(iters.max()), "roofline_frac": float(flops / (ms.sum() * 1e-3) / PEAK_FP64),
            "two_stream_groups_steps_per_s": B * K / el2, "two_stream_groups_last_result_identical": same,
            "what": "SURVEY 8(d) closed receding-horizon sequence (state advanced with the MPC's own prediction), "
                    "%d timed calls after %d warm-up calls, inputs replayed from HBM" % (K, W)}


def device_resident_loop(sb, B, N, N_gait, dev, iters=40, k_mpc=10, multiprocessing=False, loop_cus=None, groups=1, free_running=False,
                         stagger=False, on_loop_stream=False):
    """Secondary figure: the reference's own 1:10 MPC:WBC ratio, whole Controller.compute iterations
    (scripts/Controller.py:200-326) on the device — updateState, the four planners, one MPC solve every k_mpc
    iterations, WBC target assembly, InvKin + QPWBC, result + security check — nothing leaving HBM.
    multiprocessing=True: the reference's asynchronous MPC mode (scripts/MPC_Wrapper.py:150-298) as two
    compute-unit-masked streams.  Reports the free-running rate and the iteration latency when paced at dt_wbc = 2 ms.
    groups = 2: the fleet as two stream groups (Controller_groups), results identical; free_running = True additionally
    keeps the stand-in for the robots (perfect tracking of the PD targets) per group on the group's stream, so that the
    groups are never joined and one group's straggling solves run beside the other group's iterations.
    stagger = True: group g starts g * k_mpc / groups ticks late (Controller_groups), so the groups solve on different ticks."""
    import torch

    from Controller import Controller_batch

    q_init = np.array([0.0, 0.7, -1.4, -0.0, 0.7, -1.4, 0.0, -0.7, +1.4, -0.0, -0.7, +1.4])
    # the masked streams synchronise with the legacy default stream: keep the caller's own work off it
    with torch.cuda.stream(torch.cuda.Stream(dev)):
        ctl = Controller_batch(B, q_init, dt_wbc=0.002, dt_mpc=0.02, k_mpc=k_mpc, T_gait=0.02 * N, T_mpc=0.02 * N,
                               N_gait=N_gait, device=dev.index or 0, multiprocessing=multiprocessing, loop_cus=loop_cus,
                               groups=groups, stagger=(None if groups is None else stagger))
        groups, stagger = getattr(ctl, "G", 1), getattr(ctl, "stagger", False)  # (groups=None: what the object chose by itself)
        # on_loop_stream (asynchronous mode): the caller's own work (the stand-in for the robots below) runs on the control loop's
        # stream, which saves compute() the hand-over between two streams (Controller_batch.loop_stream)
        import contextlib
        ctx = torch.cuda.stream(ctl.loop_stream) if (on_loop_stream and multiprocessing and groups == 1) else contextlib.nullcontext()
        ctx.__enter__()
        # half the joystick range of the headline workload: at up to 1.5 m/s a sixth of the instances run into the
        # controller's joint-limit / torque security stop within 100 iterations (reference behaviour), which would
        # make the figure depend on how many robots have already been stopped
        vref = torch.from_numpy(np.ascontiguousarray(0.5 * sb.vref)).to(dev)
        qf = torch.zeros((B, 19), dtype=torch.float64, device=dev)
        qf[:, 2], qf[:, 6] = 0.2229, 1.0
        qf[:, 7:] = torch.from_numpy(q_init).to(dev)
        vf = torch.zeros((B, 18), dtype=torch.float64, device=dev)
        vf[:, :6] = vref
        rpy = torch.zeros((B, 3), dtype=torch.float64, device=dev)
        vs = torch.zeros((B, 12), dtype=torch.float64, device=dev)

        host_us = []  # host time of the compute() calls alone (enqueue, no synchronisation), with whether the iteration solved
        gv = [tuple(a[ctl.slice_of(g)] for a in (vref, qf, vf, rpy, vs)) for g in range(groups)] if (free_running and groups > 1) else None

        def it():
            if free_running and groups > 1:
                for g in range(groups):
                    sl = ctl.slice_of(g)
                    with torch.cuda.stream(ctl.stream_of(g)):
                        r = ctl.compute_group(g, *gv[g])
                        if ctl.group_started(g):
                            qf[sl, 7:].copy_(r.q_des)
                            vf[sl, 6:].copy_(r.v_des)
                return
            solving = any(((ctl.k - dly) % k_mpc) == 0 for dly in getattr(ctl, "_delay", [0]))
            t0 = time.perf_counter()
            r = ctl.compute(vref, qf, vf, rpy, vs)
            host_us.append((1e6 * (time.perf_counter() - t0), solving))
            if groups > 1 and not all(ctl.group_started(g) for g in range(groups)):
                for g in range(groups):  # a staggered group that has not started holds q_init: nothing to feed back yet
                    if ctl.group_started(g):
                        sl = ctl.slice_of(g)
                        qf[sl, 7:].copy_(r.q_des[sl])
                        vf[sl, 6:].copy_(r.v_des[sl])
                return
            qf[:, 7:].copy_(r.q_des)  # perfect tracking of the PD targets stands in for the robot
            vf[:, 6:].copy_(r.v_des)

        def wait_iteration():
            if free_running and groups > 1:
                for g in range(groups):
                    ctl.stream_of(g).synchronize()
            else:
                torch.cuda.current_stream().synchronize()

        for k in range(2 * k_mpc):
            it()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(iters):
            it()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        # paced like the real loop: one iteration every dt_wbc, latency = call until its PD targets are ready
        lat, nxt = [], time.perf_counter()
        for k in range(iters):
            while time.perf_counter() < nxt:
                pass
            nxt = max(nxt + 0.002, time.perf_counter())
            a = time.perf_counter()
            it()
            wait_iteration()
            lat.append(time.perf_counter() - a)
        bad = int((ctl.error_flag != 0).sum().item())
        ctx.__exit__(None, None, None)
        ctl.stop_parallel_loop()
    lat = 1e3 * np.array(lat)
    ns = np.array([t for t, sv in host_us[2 * k_mpc:] if not sv])
    what = ("whole Controller.compute iterations (state update, planners, glue, WBC every iteration, MPC every %d-th), "
            "device-resident, batch %d, reference velocities = half the headline workload's" % (k_mpc, B))
    if groups > 1:
        what += "; fleet as %d stream groups (%s)" % (groups, "never joined: the robots' stand-in runs per group" if free_running
                                                      else "joined on the caller's stream every iteration")
        if stagger:
            what += "; staggered: group g started g * k_mpc / groups ticks late, the groups' solves fall on different ticks"
    if multiprocessing:
        what += ("; asynchronous MPC mode: solves on their own stream (224 compute units), the control loop on a stream "
                 "with the other 32, a result adopted when its event has completed")
    return {"value": B * iters / el, "unit": "control iterations/s", "iterations": iters, "k_mpc": k_mpc,
            "ms_per_iteration": 1e3 * el / iters, "paced_2ms_latency_ms": {"median": float(np.median(lat)), "worst": float(lat.max())},
            "host_us_per_nonsolving_compute": ({"median": float(np.median(ns)), "p90": float(np.percentile(ns, 90))} if len(ns) else None),
            "instances_in_security_stop": bad, "what": what}


def realtime_slot_leg(sb_full, N, N_gait, dev, batches=(64, 256, 1024, 2048, 4096)):
    """How many robots fit the reference's real-time slot: the 1:10 loop (scripts/Controller.py:246, dt_wbc = 2 ms,
    src/config_solo12.yaml:6) paced at 2 ms for a growing fleet, in the synchronous mode (the iteration that solves carries the
    whole MPC launch; this is the default object), as two / four staggered stream groups (opt-in, what Controller_batch.for_deadline
    builds between 1025 and 2048 robots) and in the asynchronous
    mode (the solve on its own compute units; the caller on the loop's stream): median / worst time from the call of compute() to
    its PD targets being ready, host time of a compute() that does not solve, and the largest fleet whose WORST iteration stays
    inside the slot."""
    import synth

    rows = []
    for B in batches:
        sb = sb_full if B == sb_full.B else synth.SyntheticBatch(B, N, N_gait=N_gait, gaits=("trot",), n_seq=1)
        row = {"batch": B}
        for name, kw in (("sync", dict(groups=1)), ("two_staggered_groups", dict(groups=2, stagger=True)),
                         ("four_staggered_groups", dict(groups=4, stagger=True)),
                         ("async", dict(groups=1, multiprocessing=True, on_loop_stream=True))):
            if name == "two_staggered_groups" and B < 2048:
                continue  # (one handle fits the slot there)
            if name == "four_staggered_groups" and B < 4096:
                continue  # (only where two groups no longer fit the slot)
            r = device_resident_loop(sb, B, N, N_gait, dev, **kw)
            row[name] = {"paced_median_ms": r["paced_2ms_latency_ms"]["median"], "paced_worst_ms": r["paced_2ms_latency_ms"]["worst"],
                         "host_us_per_nonsolving_compute": r["host_us_per_nonsolving_compute"],
                         "free_running_M_iterations_per_s": r["value"] / 1e6, "stopped": r["instances_in_security_stop"]}
        rows.append(row)
    fits = {}
    for name in ("sync", "two_staggered_groups", "four_staggered_groups", "async"):
        ok = [r["batch"] for r in rows if name in r and r[name]["paced_worst_ms"] < 2.0]
        fits[name] = max(ok) if ok else None
    import Controller

    fits["default_object"] = fits["sync"]  # Controller_batch(B, ...) without arguments is the single handle at every fleet size
    return {"slot_ms": 2.0, "k_mpc": 10, "rows": rows, "largest_batch_with_worst_iteration_inside_the_slot": fits,
            "table_shipped_in_Controller.recommended_mode": dict(Controller.REALTIME_SLOT_FITS),
            "what": "1:10 control loop paced at dt_wbc = 2 ms, 40 paced iterations (4 of them solve) after 20 warm-up and 40 free-running "
                    "ones -- the loop's first two solves (QP set-up, cold start: about twice as many ADMM iterations) are not in the figure, "
                    "as the reference's first iterations would not be; latency = compute() call until the iteration's PD targets are ready (incl. the two copies that stand in for "
                    "the robots); batches tested: %s" % (list(batches),)}


def physical_cores_allowed():
    """Physical cores among the logical CPUs this process may run on ((physical id, core id) pairs of /proc/cpuinfo); the
    number of allowed logical CPUs if /proc/cpuinfo does not say."""
    allowed = os.sched_getaffinity(0)
    cores, proc, phys = set(), None, None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("processor"):
                proc = int(ln.split(":")[1])
            elif ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id") and proc in allowed:
                cores.add((phys, ln.split(":")[1].strip()))
    except Exception:
        pass
    return len(cores) if cores else len(allowed)


def host_cpu_info():
    """What the CPU baseline ran on: model name, logical CPUs of the box and of this process, clocks from /proc/cpuinfo."""
    info = {"model": None, "logical_cpus": os.cpu_count(), "cpus_allowed": ALLOWED_CPUS, "mhz_now_max": None,
            "mhz_max": None}
    try:
        mhz = []
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and info["model"] is None:
                info["model"] = ln.split(":", 1)[1].strip()
            elif ln.startswith("cpu MHz"):
                mhz.append(float(ln.split(":", 1)[1]))
        if mhz:
            info["mhz_now_max"] = max(mhz)
    except Exception:
        pass
    try:
        info["mhz_max"] = int(open("/sys/devices/system/cpu/cpu0/cpufreq/cpuinfo_max_freq").read()) / 1e3
    except Exception:
        pass
    try:
        info["loadavg_1min"] = os.getloadavg()[0]
    except Exception:
        pass
    return info


def omp_pinned():
    """True when the OpenMP runtime was told to bind the oracle's threads to places (OMP_PROC_BIND in the caller's environment).
    bench.py does not set it: binding also narrows the initial thread's CPU mask, which every helper thread of the process --
    the HIP runtime's included -- inherits; measured once by accident, the CPU leg then ran on the 2 hardware threads of one
    core.  Unpinned threads on a shared host are part of the round-to-round spread of this figure."""
    v = os.environ.get("OMP_PROC_BIND", "").lower()
    return v not in ("", "false", "0")


def cpu_baseline(synth, Bc, N, N_gait, gaits, threads, steps):
    """CPU restatement (oracle/, 'port') on a bounded sample of the same workload: Bc instances x `steps` control steps
    of the headline's open-loop sequence (the first one, which sets the QP up, is excluded like the GPU warm-up), one
    instance per OpenMP thread over the box's host cores.  Returns the figure and the per-step outputs (for the
    accuracy leg)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    oracle.build(fast=True)
    cores = max(1, min(int(threads), ALLOWED_CPUS))
    sb = synth.SyntheticBatch(Bc, N, N_gait=N_gait, gaits=gaits, n_seq=steps + 1)
    mpc = oracle.MPCBatch(Bc, 0.02, N, 0.02 * N, N_gait, fast=True)
    wbc = oracle.WbcBatch(Bc, 0.002, fast=True)
    t_mpc = t_wbc = 0.0
    outs = []
    for s in range(steps + 1):
        d = sb.step(s)
        a = time.perf_counter()
        r = mpc.run(s, d["xref"], d["fsteps"], cores)
        b = time.perf_counter()
        tau, _, _, f = wbc.compute(d["q"], d["dq"], np.ascontiguousarray(r[:, 12:, 0]), d["contacts"], d["pgoals"],
                                   d["vgoals"], d["agoals"], cores)
        c = time.perf_counter()
        outs.append((r, tau, f))
        if s > 0:
            t_mpc += b - a
            t_wbc += c - b
    tot = t_mpc + t_wbc
    # BASELINE config 1: a single robot on a single thread (the reference's own deployment shape); best of three passes
    # (the figure moved by 23 % between two rounds' boxes on unchanged code: a shared host)
    sb1 = synth.SyntheticBatch(1, N, N_gait=N_gait, gaits=gaits, n_seq=33)
    seq1 = [sb1.step(s) for s in range(33)]
    t1s = []
    for _ in range(3):
        m1, w1 = oracle.MPCBatch(1, 0.02, N, 0.02 * N, N_gait, fast=True), oracle.WbcBatch(1, 0.002, fast=True)
        t1 = 0.0
        for s, d in enumerate(seq1):
            a = time.perf_counter()
            r = m1.run(s, d["xref"], d["fsteps"], 1)
            w1.compute(d["q"], d["dq"], np.ascontiguousarray(r[:, 12:, 0]), d["contacts"], d["pgoals"], d["vgoals"], d["agoals"], 1)
            if s > 0:
                t1 += time.perf_counter() - a
        t1s.append(t1)
    t1 = min(t1s)
    # the same restatement on ALL the host cores this process may use (north_star: "the same box's host cores"): one thread per
    # physical core, the sample grown in proportion so that the leg takes the same wall time as the 16-thread one
    t_all = min(ALLOWED_CPUS, physical_cores_allowed())
    if t_all > cores:
        # (bounded: at most four times the 16-thread sample and ~25 s of wall time -- on a shared host the per-thread rate of 128
        # threads is a fraction of that of 16, BENCH r6: the leg must not stretch the default bench run)
        B_all = Bc * min(t_all // cores, 4)
        sba = synth.SyntheticBatch(B_all, N, N_gait=N_gait, gaits=gaits, n_seq=steps + 1)
        ma, wa = oracle.MPCBatch(B_all, 0.02, N, 0.02 * N, N_gait, fast=True), oracle.WbcBatch(B_all, 0.002, fast=True)
        ta, done = 0.0, 0
        for s in range(steps + 1):
            d = sba.step(s)
            a = time.perf_counter()
            r = ma.run(s, d["xref"], d["fsteps"], t_all)
            wa.compute(d["q"], d["dq"], np.ascontiguousarray(r[:, 12:, 0]), d["contacts"], d["pgoals"], d["vgoals"], d["agoals"], t_all)
            if s > 0:
                ta += time.perf_counter() - a
                done += 1
                if ta > 25.0 and done >= 3:
                    break
        all_cores = {"value": B_all * done / ta, "unit": "steps/s", "cores": t_all, "kind": "port",
                     "loadavg_1min_after": (os.getloadavg()[0] if hasattr(os, "getloadavg") else None),
                     "sample": "%d instances x %d control steps (after the set-up step) of the headline sequence, %.1f s of wall time; one "
                               "OpenMP thread per physical core this process may use (%d logical CPUs allowed, not pinned, host shared "
                               "with other jobs: see host_cpu.loadavg_1min); CPU restatement oracle/ (OSQP-0.6-style, not OSQP itself)"
                               % (B_all, done, ta, ALLOWED_CPUS)}
        del ma, wa, sba
    else:
        all_cores = {"value": None, "cores": t_all, "sample": "not run: this process may use %d logical CPUs / %d physical cores, no more "
                                                             "than the %d threads of the figure above" % (ALLOWED_CPUS, t_all, cores)}
    return ({"value": Bc * steps / tot, "unit": "steps/s", "cores": cores, "kind": "port", "all_cores": all_cores,
             "host_cpu": host_cpu_info(), "pinned": omp_pinned(),
             "single_instance_single_thread_steps_per_s": 32 / t1,
             "single_instance_single_thread_passes_steps_per_s": [32 / t for t in t1s],
             "sample": "%d instances x %d control steps (after the set-up step) of the headline sequence, %.1f s of CPU work; "
                       "CPU restatement oracle/ (OSQP-0.6-style, not OSQP itself), gcc -O3 -march=native, OpenMP one instance "
                       "per thread" % (Bc, steps, tot),
             "mpc_s": t_mpc, "wbc_s": t_wbc}, outs)


def accuracy_vs_oracle(synth, ref_out, Bc, N, N_gait, gaits, steps, dev):
    """The metric's accuracy half: the HIP path on the cpu_baseline sample (same seeds, same sequence, every step incl.
    the set-up call), compared with the oracle's outputs.  north_star's bar: 1e-4 relative."""
    import torch

    import qrw_hip

    sb = synth.SyntheticBatch(Bc, N, N_gait=N_gait, gaits=gaits, n_seq=steps + 1)
    eng = qrw_hip.Batch(Bc, n_steps=N, N_gait=N_gait, dt_mpc=0.02, T_gait=0.02 * N, dt_wbc=0.002, device=dev.index or 0)
    tau_abs = tau_ref_max = f_abs = f_ref_max = x_abs = x_ref_max = 0.0
    for s in range(steps + 1):
        d = sb.step(s)
        t = {k: torch.from_numpy(np.ascontiguousarray(d[k])).to(dev) for k in
             ("xref", "fsteps", "q", "dq", "contacts", "pgoals", "vgoals", "agoals")}
        o = eng.mpc_solve(t["xref"], t["fsteps"], s)
        w = eng.wbc_compute(t["q"], t["dq"], o[:, 12:, 0].contiguous(), t["contacts"], t["pgoals"], t["vgoals"], t["agoals"])
        r, tau, f = ref_out[s]
        og, tg, fg = o.cpu().numpy(), w["tau_ff"].cpu().numpy(), w["f_with_delta"].cpu().numpy()
        tau_abs, tau_ref_max = max(tau_abs, np.abs(tg - tau).max()), max(tau_ref_max, np.abs(tau).max())
        f_abs, f_ref_max = max(f_abs, np.abs(fg - f).max()), max(f_ref_max, np.abs(f).max())
        x_abs, x_ref_max = max(x_abs, np.abs(og - r).max()), max(x_ref_max, np.abs(r).max())
    eng.close()
    return {"torque_max_abs_err": float(tau_abs), "torque_max_rel_err": float(tau_abs / tau_ref_max),
            "force_max_rel_err": float(f_abs / f_ref_max), "mpc_result_max_rel_err": float(x_abs / x_ref_max),
            "torque_scale_Nm": float(tau_ref_max), "tolerance_rel": 1e-4,
            "within_tolerance": bool(max(tau_abs / tau_ref_max, f_abs / f_ref_max, x_abs / x_ref_max) < 1e-4),
            "sample": "%d instances x %d control steps, GPU (strict build) vs CPU oracle (-O3 -march=native build used for the "
                      "timing), max over all instances and steps; relative = max abs error / max abs reference" % (Bc, steps + 1)}


if __name__ == "__main__":
    main()
