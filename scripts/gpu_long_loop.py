"""Soak of the device-resident control loop: Controller_batch (batch 4096, 1:10, perfect tracking of the PD targets in
place of a robot) for thousands of iterations with the joystick velocity and the gait changing on the way.  Reports the
security stops, the solver statuses seen and whether anything non-finite ever appeared.
python scripts/gpu_long_loop.py [iterations] [async]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np, torch
from Controller import Controller_batch

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
mp = len(sys.argv) > 2
B, N, dev = 4096, int(os.environ.get("QRW_LOOP_N", "16")), torch.device("cuda", 0)
rng = np.random.default_rng(11)
q_init = np.array([0.0, 0.7, -1.4, -0.0, 0.7, -1.4, 0.0, -0.7, +1.4, -0.0, -0.7, +1.4])


def new_vref(scale):
    v = np.zeros((B, 6))
    v[:, 0] = rng.uniform(-0.4, 0.8, B) * scale
    v[:, 1] = rng.uniform(-0.3, 0.3, B) * scale
    v[:, 5] = rng.uniform(-0.6, 0.6, B) * scale
    v[::17, 5] = 0.0  # the yaw-rate == 0 branches of the planners
    return torch.from_numpy(v).to(dev)


with torch.cuda.stream(torch.cuda.Stream(dev)):
    # QRW_LOOP_GROUPS=auto: two staggered stream groups (what Controller_batch built by itself from 2048 robots on in rounds 4-5)
    grp = os.environ.get("QRW_LOOP_GROUPS", "1")
    ctl = Controller_batch(B, q_init, groups=(2 if grp == "auto" else int(grp)), stagger=(grp == "auto"), multiprocessing=mp, T_gait=0.02 * N, T_mpc=0.02 * N,
                           N_gait=max(20, N + 4))
    vref = new_vref(0.5)
    qf = torch.zeros((B, 19), dtype=torch.float64, device=dev); qf[:, 2], qf[:, 6] = 0.2229, 1.0
    qf[:, 7:] = torch.from_numpy(q_init).to(dev)
    vf = torch.zeros((B, 18), dtype=torch.float64, device=dev); vf[:, :6] = vref
    rpy = torch.zeros((B, 3), dtype=torch.float64, device=dev); vs = torch.zeros((B, 12), dtype=torch.float64, device=dev)
    statuses, max_it, nonfinite, t0 = {}, 0, 0, time.perf_counter()
    for k in range(iters):
        code = 0
        if k and k % 600 == 0:
            vref = new_vref(0.25 + 0.5 * rng.random()); vf[:, :6] = vref
            code = int(rng.integers(1, 4))  # trot / pacing / bounding families of the reference's joystick codes
        r = ctl.compute(vref, qf, vf, rpy, vs, joystick_code=code)
        qf[:, 7:].copy_(r.q_des); vf[:, 6:].copy_(r.v_des)
        if k % 100 == 99:
            torch.cuda.synchronize()
            stop = ctl.error_flag != 0
            live = ~stop
            nonfinite += int((~torch.isfinite(r.tau_ff[live])).sum().item()) + int((~torch.isfinite(r.q_des[live])).sum().item())
            st = ctl.stats()["mpc"]
            for s_ in np.unique(st["status"]): statuses[int(s_)] = statuses.get(int(s_), 0) + int((st["status"] == s_).sum())
            max_it = max(max_it, int(st["iters"].max()))
            if k % 500 == 499:
                print("iteration %5d: %4d of %d instances in security stop, MPC statuses so far %s, max ADMM iterations %d, non-finite outputs of running robots %d"
                      % (k + 1, int(stop.sum().item()), B, statuses, max_it, nonfinite), flush=True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ctl.stop_parallel_loop()
print("LONG LOOP %s: %d iterations x %d robots in %.2f s (%.2f M iterations/s incl. the checks), security stops %d, non-finite %d, statuses %s"
      % ("async" if mp else "sync", iters, B, el, B * iters / el / 1e6, int((ctl.error_flag != 0).sum().item()), nonfinite, statuses))
