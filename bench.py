#!/usr/bin/env python3
"""bench.py — control-steps/sec (MPC + WBC) of the MI355X hot path on synthetic Solo12 trot states.

One "step" = one MPC::run + one wbc_controller.compute for every instance of the batch
(ratio 1:1).  Default workload = BASELINE.json configs[2]: batch 4096, N = 16,
trot, MPC + QPWBC + InvKin on one MI355X.  With --gpus N every rank owns its own 4096 instances
(weak scaling).  Instances are independent, so the step has no collective; --gather-results adds the
optional all-gather of the packed results (RCCL over xGMI) a central logger would want.

Prints ONE JSON line (rank 0): metric/value/unit/... plus
  roofline     — dominant kernel (mpc_solve_kernel): algorithmic FP64 flops per launch (measured ADMM
                 iteration counts x SURVEY §8(d) per-iteration figure) / its average duration measured
                 with HIP events on the launch stream, against the gfx950 FP64 peak;
  cpu_baseline — the CPU oracle (oracle/, "port", -O3 -march=native, OpenMP) timed on a bounded sample
                 of the same workload on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

# algorithmic work per unit (SURVEY.md §8(d), N = 16; DESIGN.md restates them)
F_ITER = 64.8e3      # flops per ADMM iteration of one MPC instance
F_FAC = 0.33e6       # flops per KKT factorisation
F_ASM = 3.0e3        # assembly
F_WBC = 15.0e3       # kinematics + 2x RNEA + InvKin + QP build
F_WBC_IT = 1.0e3     # per 12-variable ADMM iteration
B_ALG = 36.4e3       # compulsory bytes per control step
PEAK_FP64 = 78.6e12  # gfx950 FP64 vector = matrix peak
PEAK_HBM = 8.0e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--n-steps", type=int, default=16, help="MPC horizon")
    ap.add_argument("--gaits", type=str, default="trot")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather-results", action="store_true",
                    help="N > 1: all-gather every rank's packed results each step (not part of the control path)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary closed-loop figures (profiling runs)")
    ap.add_argument("--cpu-sample", type=int, default=256, help="instances in the CPU baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads for the CPU baseline (a 1-GPU box owns 16 cores)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # rehearsal knobs (1-GPU box): QRW_SINGLE_DEVICE=1 puts every rank on cuda:0, QRW_DIST_BACKEND=gloo avoids RCCL
    if os.environ.get("QRW_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("QRW_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import qrw_hip
    import synth
    from sharding import ResultGatherer, pack_results

    B, N = args.batch, args.n_steps
    N_gait = max(20, N + 4)
    W, K = max(args.warmup, 1), args.steps  # the first call (num_iter == 0) is the QP setup: always untimed
    gaits = tuple(args.gaits.split(","))
    sb = synth.SyntheticBatch(B, N, N_gait=N_gait, gaits=gaits, n_seq=W + K, b0=rank * B)
    t_gen = time.time()
    steps = [sb.step(s) for s in range(W + K)]
    t_gen = time.time() - t_gen

    def dev_t(key):
        return [torch.from_numpy(np.ascontiguousarray(st[key])).to(dev) for st in steps]

    xref, fsteps = dev_t("xref"), dev_t("fsteps")
    q, dq, contacts = dev_t("q"), dev_t("dq"), dev_t("contacts")
    pg, vg, ag = dev_t("pgoals"), dev_t("vgoals"), dev_t("agoals")

    eng = qrw_hip.Batch(B, n_steps=N, N_gait=N_gait, dt_mpc=0.02, T_gait=0.02 * N, dt_wbc=0.002, device=local_rank)
    mpc_out = torch.empty((B, 24, N), dtype=torch.float64, device=dev)
    f_cmd = torch.empty((B, 12), dtype=torch.float64, device=dev)
    wbc_out = None
    gather = ResultGatherer(B, 48, dev) if (world > 1 and args.gather_results) else None
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    ev_w = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    it_mpc, it_wbc = [], []

    def one_step(s, timed_idx=None):
        nonlocal wbc_out
        if timed_idx is not None:
            ev[timed_idx][0].record()
        eng.mpc_solve(xref[s], fsteps[s], s, out=mpc_out)
        if timed_idx is not None:
            ev[timed_idx][1].record()
        f_cmd.copy_(mpc_out[:, 12:, 0])
        if timed_idx is not None:
            ev_w[timed_idx][0].record()
        wbc_out = eng.wbc_compute(q[s], dq[s], f_cmd, contacts[s], pg[s], vg[s], ag[s], out=wbc_out)
        if timed_idx is not None:
            ev_w[timed_idx][1].record()
        if gather is not None:
            gather.gather(pack_results(wbc_out["tau_ff"], wbc_out["f_with_delta"], wbc_out["qdes"], wbc_out["vdes"]))

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(W):
        one_step(s)
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        one_step(W + i, i)
        # iteration counts are read back AFTER the timed region (they stay on the device)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # per-kernel figures (rank 0): durations from the HIP events, iteration counts of the LAST step
    mpc_ms = np.array([a.elapsed_time(b) for a, b in ev])
    wbc_ms = np.array([a.elapsed_time(b) for a, b in ev_w])
    ms = eng.mpc_stats()
    ws = eng.wbc_stats()
    n_ok = int((ms["status"] == 1).sum())
    # iteration counts differ per step; replay the stats of the last step as representative and
    # scale the flops of every timed launch by its own duration share
    flops_launch = float(ms["iters"].astype(np.float64).sum() * F_ITER + B * (F_FAC + F_ASM))
    dur = float(mpc_ms[-1]) * 1e-3
    achieved = flops_launch / dur
    total_steps = world * B * K
    value = total_steps / elapsed

    out = {
        "metric": "control-steps/sec (MPC+WBC)", "value": value, "unit": "steps/s", "n_gpus": world,
        "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "Solo12 %s, batch %d per GPU, horizon N=%d, MPC (OSQP-style ADMM) + WBC (InvKin + "
                               "RNEA + box-QP) per control step, ratio 1:1" % ("/".join(gaits), B, N),
                   "batch_per_gpu": B, "n_steps": N, "gaits": list(gaits), "parallelism": "batch-sharded x%d" % world},
        "roofline": {"kernel": "mpc_solve_kernel", "bound": "mfma", "achieved": achieved / 1e12, "peak": PEAK_FP64 / 1e12,
                     "unit": "TFLOP/s", "frac": achieved / PEAK_FP64, "traffic": None,
                     "launch_ms": float(mpc_ms[-1]), "launch_ms_mean": float(mpc_ms.mean()),
                     "mean_admm_iters": float(ms["iters"].mean()), "max_admm_iters": int(ms["iters"].max()),
                     "hbm_frac_algorithmic": (B * B_ALG / dur) / PEAK_HBM,
                     "note": "compute roof: FP64 peak of gfx950 (vector = matrix = 78.6 TFLOP/s); the kernel is issue / "
                             "latency bound and issues no MFMA (its sweeps run on the FP64 VALU with DPP, DESIGN.md 4.1)"},
        "kernels_ms": {"mpc_solve_kernel": float(mpc_ms.mean()), "wbc_kernel": float(wbc_ms.mean())},
        "solver": {"mpc_solved": n_ok, "mpc_instances": B, "wbc_mean_iters": float(ws["iters"].mean())},
        "mpc_solves_per_s": world * B * K / (mpc_ms.sum() * 1e-3) if world == 1 else None,
        "wbc_steps_per_s": world * B * K / (wbc_ms.sum() * 1e-3) if world == 1 else None,
        "input_gen_s": t_gen,
    }

    if rank == 0 and world == 1:
        out["roofline"]["traffic"] = pmc_traffic_bytes()
    if rank == 0 and world == 1 and not args.no_secondary:
        out["secondary_ratio_1_10"] = device_resident_loop(sb, B, N, N_gait, dev)
        out["secondary_ratio_1_10_async"] = device_resident_loop(sb, B, N, N_gait, dev, multiprocessing=True)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(synth, args.cpu_sample, N, N_gait, gaits, args.cpu_threads)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def pmc_traffic_bytes():
    """HBM bytes per mpc_solve_kernel launch from the committed rocprofv3 PMC passes (profiles/, FETCH_SIZE and
    WRITE_SIZE are reported in KiB; 8-byte-per-lane accesses are uncalibrated on gfx950)."""
    path = os.path.join(ROOT, "profiles", "r1_pmc_summary_bench_b4096.json")
    try:
        rows = json.load(open(path))
        tot = 0.0
        for r in rows:
            if "mpc_solve_kernel" in r["kernel"] and r["counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
                tot += r["mean"] * 1024.0
        return tot or None
    except Exception:
        return None


def device_resident_loop(sb, B, N, N_gait, dev, iters=40, k_mpc=10, multiprocessing=False):
    """Secondary figure: the reference's own 1:10 MPC:WBC ratio, whole Controller.compute iterations
    (scripts/Controller.py:200-326) on the device — updateState, the four planners, one MPC solve every k_mpc
    iterations, WBC target assembly, InvKin + QPWBC, result + security check — nothing leaving HBM.
    multiprocessing=True: the reference's asynchronous MPC mode (scripts/MPC_Wrapper.py:150-298) as two
    compute-unit-masked streams.  Reports the free-running rate and the iteration latency when paced at dt_wbc = 2 ms."""
    from Controller import Controller_batch

    q_init = np.array([0.0, 0.7, -1.4, -0.0, 0.7, -1.4, 0.0, -0.7, +1.4, -0.0, -0.7, +1.4])
    # the masked streams synchronise with the legacy default stream: keep the caller's own work off it
    with torch.cuda.stream(torch.cuda.Stream(dev)):
        ctl = Controller_batch(B, q_init, dt_wbc=0.002, dt_mpc=0.02, k_mpc=k_mpc, T_gait=0.02 * N, T_mpc=0.02 * N,
                               N_gait=N_gait, device=dev.index or 0, multiprocessing=multiprocessing)
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

        def it():
            r = ctl.compute(vref, qf, vf, rpy, vs)
            qf[:, 7:].copy_(r.q_des)  # perfect tracking of the PD targets stands in for the robot
            vf[:, 6:].copy_(r.v_des)

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
            torch.cuda.current_stream().synchronize()
            lat.append(time.perf_counter() - a)
        bad = int((ctl.error_flag != 0).sum().item())
        ctl.stop_parallel_loop()
    lat = 1e3 * np.array(lat)
    what = ("whole Controller.compute iterations (state update, planners, glue, WBC every iteration, MPC every %d-th), "
            "device-resident, batch %d, reference velocities = half the headline workload's" % (k_mpc, B))
    if multiprocessing:
        what += ("; asynchronous MPC mode: solves on their own stream (224 compute units), the control loop on a stream "
                 "with the other 32, a result adopted when its event has completed")
    return {"value": B * iters / el, "unit": "control iterations/s", "iterations": iters, "k_mpc": k_mpc,
            "ms_per_iteration": 1e3 * el / iters, "paced_2ms_latency_ms": {"median": float(np.median(lat)), "worst": float(lat.max())},
            "instances_in_security_stop": bad, "what": what}


def cpu_baseline(synth, Bc, N, N_gait, gaits, threads, steps=8):
    """CPU restatement (oracle/, 'port') on a bounded sample of the same workload: Bc instances x `steps`
    receding-horizon control steps (the first one, which sets the QP up, is excluded like the GPU warm-up),
    one instance per OpenMP thread over all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    oracle.build(fast=True)
    cores = max(1, min(int(threads), len(os.sched_getaffinity(0))))
    sb = synth.SyntheticBatch(Bc, N, N_gait=N_gait, gaits=gaits, n_seq=steps + 1)
    mpc = oracle.MPCBatch(Bc, 0.02, N, 0.02 * N, N_gait, fast=True)
    wbc = oracle.WbcBatch(Bc, 0.002, fast=True)
    t_mpc = t_wbc = 0.0
    for s in range(steps + 1):
        d = sb.step(s)
        a = time.perf_counter()
        r = mpc.run(s, d["xref"], d["fsteps"], cores)
        b = time.perf_counter()
        wbc.compute(d["q"], d["dq"], np.ascontiguousarray(r[:, 12:, 0]), d["contacts"], d["pgoals"], d["vgoals"],
                    d["agoals"], cores)
        c = time.perf_counter()
        if s > 0:
            t_mpc += b - a
            t_wbc += c - b
    tot = t_mpc + t_wbc
    # BASELINE config 1: a single robot on a single thread (the reference's own deployment shape)
    sb1 = synth.SyntheticBatch(1, N, N_gait=N_gait, gaits=gaits, n_seq=33)
    m1, w1 = oracle.MPCBatch(1, 0.02, N, 0.02 * N, N_gait, fast=True), oracle.WbcBatch(1, 0.002, fast=True)
    t1 = 0.0
    for s in range(33):
        d = sb1.step(s)
        a = time.perf_counter()
        r = m1.run(s, d["xref"], d["fsteps"], 1)
        w1.compute(d["q"], d["dq"], np.ascontiguousarray(r[:, 12:, 0]), d["contacts"], d["pgoals"], d["vgoals"], d["agoals"], 1)
        if s > 0:
            t1 += time.perf_counter() - a
    return {"value": Bc * steps / tot, "unit": "steps/s", "cores": cores, "kind": "port",
            "single_instance_single_thread_steps_per_s": 32 / t1,
            "sample": "%d instances x %d control steps (after the set-up step), CPU restatement oracle/ "
                      "(OSQP-0.6-style, not OSQP itself), gcc -O3 -march=native, OpenMP one instance per thread" % (Bc, steps),
            "mpc_s": t_mpc, "wbc_s": t_wbc}


if __name__ == "__main__":
    main()
