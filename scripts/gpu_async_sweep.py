"""Control-iteration time of Controller_batch (batch 4096) in synchronous mode and in the asynchronous MPC mode for
several compute-unit splits between the control loop's stream and the MPC's stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "quadruped-reactive-walking_amd")]
import numpy as np, torch
from Controller import Controller_batch

B, iters = 4096, 60
PERIOD = 0.0
dev = torch.device("cuda:0")
q_init = np.array([0.0, 0.7, -1.4, -0.0, 0.7, -1.4, 0.0, -0.7, +1.4, -0.0, -0.7, +1.4])
rng = np.random.default_rng(1)
vref_h = np.zeros((B, 6)); vref_h[:, 0] = rng.uniform(-0.3, 0.8, B); vref_h[:, 1] = rng.uniform(-0.3, 0.3, B); vref_h[:, 5] = rng.uniform(-0.5, 0.5, B)

def run(**kw):
    # CU-masked streams are "blocking" streams (they synchronise with the legacy default stream), so the caller's own
    # work must not sit on the default stream or everything serialises again
    with torch.cuda.stream(torch.cuda.Stream(dev)):
        return run_(**kw)


def run_(**kw):
    ctl = Controller_batch(B, q_init, groups=1, **kw)
    vref = torch.from_numpy(vref_h).to(dev)
    qf = torch.zeros((B, 19), dtype=torch.float64, device=dev); qf[:, 2], qf[:, 6] = 0.2229, 1.0
    qf[:, 7:] = torch.from_numpy(q_init).to(dev)
    vf = torch.zeros((B, 18), dtype=torch.float64, device=dev); vf[:, :6] = vref
    rpy = torch.zeros((B, 3), dtype=torch.float64, device=dev); vs = torch.zeros((B, 12), dtype=torch.float64, device=dev)
    def it():
        r = ctl.compute(vref, qf, vf, rpy, vs)
        qf[:, 7:].copy_(r.q_des); vf[:, 6:].copy_(r.v_des)
    for _ in range(20): it()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lat = []
    period = kw.pop("_period", 0.0) if False else PERIOD
    nxt = time.perf_counter()
    for _ in range(iters):
        while time.perf_counter() < nxt: pass
        nxt = max(nxt + PERIOD, time.perf_counter()) if PERIOD else nxt
        a = time.perf_counter(); it(); torch.cuda.current_stream().synchronize(); lat.append(time.perf_counter() - a)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    ctl.stop_parallel_loop()
    lat = np.array(lat) * 1e3
    return 1e3 * el / iters, np.median(lat), lat.max()

for PERIOD in (0.0, 2e-3):
    print("pacing: %s" % ("free-running" if not PERIOD else "one iteration every %.1f ms (the reference's dt_wbc)" % (PERIOD * 1e3)))
    print("  sync: %.3f ms/iteration (latency median %.3f, worst %.3f)" % run())
    for lc in (16, 32, 64):
        print("  async loop_cus=%d: %.3f ms/iteration (latency median %.3f, worst %.3f)" % ((lc,) + run(multiprocessing=True, loop_cus=lc)))
