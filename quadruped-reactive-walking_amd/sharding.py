"""Batch sharding over the GPUs of one node.

Every robot instance is independent (own QP data, own solver history), so the batch is cut into
contiguous slices, one per rank, each with its own qrw handle; nothing is exchanged during the
MPC / WBC step.  The only collective is the all-gather of results (torques, forces, joint
targets: 48 doubles per instance) — RCCL over xGMI when the process group is `nccl`, and the
same code runs on `gloo` for the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Contiguous [lo, hi) slice of `total` instances owned by `rank` (remainder to the low ranks)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_results(tau_ff, f_with_delta, qdes, vdes):
    """(B,12),(B,12),(B,19),(B,18) -> (B,48): tau_ff | f | q_des[7:] | v_des[6:] (what Controller.py:306-310 consumes)."""
    return torch.cat([tau_ff, f_with_delta, qdes[:, 7:], vdes[:, 6:]], dim=1).contiguous()


class ResultGatherer:
    """All-gathers equally sized per-rank result blocks into one [world*B_local, width] tensor."""

    def __init__(self, b_local, width, device, dtype=torch.float64, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # RCCL gathers device tensors directly; other backends (gloo rehearsals on CPU or on a 1-GPU box) stage on the host
        self.host_staged = dist.is_initialized() and dist.get_backend(group) != "nccl" and torch.device(device).type != "cpu"
        self.out = torch.empty((self.world * b_local, width), dtype=dtype, device="cpu" if self.host_staged else device)

    def gather(self, local, async_op=False):
        if self.world == 1:
            self.out.copy_(local)
            return None
        if self.host_staged:
            local = local.cpu()
        return dist.all_gather_into_tensor(self.out, local, group=self.group, async_op=async_op)


class TorqueGatherPipeline:
    """BASELINE config 5's collective: every control step each rank all-gathers the joint torques of its shard
    (12 f64 per instance) so that every rank holds all robots' torques — issued on a side stream right after the step's
    WBC kernel and left in flight while the next step computes (one step deep, two result buffers).

    RCCL path (backend nccl): the collective is enqueued behind an event of the producing stream; `wait_buffer_free`
    makes the PRODUCING STREAM (not the host) wait for the gather that last read a buffer before it is overwritten.
    Other backends (gloo rehearsals on a CPU or on a 1-GPU box) stage through the host with the same call sequence."""

    def __init__(self, b_local, device, width=12, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.nccl = dist.get_backend(group) == "nccl"
        self.dev = torch.device(device)
        self.on_gpu = self.dev.type == "cuda"
        self.b_local, self.width = int(b_local), int(width)
        out_dev = self.dev if self.nccl else torch.device("cpu")
        self.out = [torch.empty((self.world * self.b_local, self.width), dtype=torch.float64, device=out_dev) for _ in range(2)]
        self.side = torch.cuda.Stream(self.dev) if (self.on_gpu and self.nccl) else None
        self.work = [None, None]
        self.last = None

    def issue(self, i, tau):
        """Start the all-gather of this step's torques `tau` (b_local, width) into result buffer i."""
        if self.nccl:
            produced = torch.cuda.Event()
            produced.record(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(self.side):
                self.side.wait_event(produced)
                self.work[i] = dist.all_gather_into_tensor(self.out[i], tau, group=self.group, async_op=True)
        else:
            local = tau.cpu() if self.on_gpu else tau.clone()
            self.work[i] = dist.all_gather_into_tensor(self.out[i], local, group=self.group, async_op=True)
        self.last = i

    def wait_buffer_free(self, i):
        """Before the producer overwrites the torque buffer that gather i read: order it behind that gather."""
        w = self.work[i]
        if w is not None:
            w.wait()  # nccl: the current (producing) stream waits on the device; gloo: the host waits
            self.work[i] = None

    def drain(self):
        for i in (0, 1):
            self.wait_buffer_free(i)

    def check_last(self, rank, world, local_tau):
        """After drain(): the last gathered buffer holds this rank's own torques in its slot, is finite, and every
        rank's slot sums to the checksum that rank computed of its own torques."""
        if self.last is None:
            return None
        if self.on_gpu:
            torch.cuda.synchronize(self.dev)
        got = self.out[self.last]
        mine = local_tau.to(got.device)
        ok = bool(torch.equal(got[rank * self.b_local:(rank + 1) * self.b_local], mine)) and bool(torch.isfinite(got).all())
        sums = torch.empty((world,), dtype=torch.float64, device=got.device)
        dist.all_gather_into_tensor(sums, mine.sum().reshape(1), group=self.group)
        block = got.reshape(world, -1).sum(dim=1)
        ok = ok and bool(torch.allclose(block, sums, rtol=1e-12, atol=1e-9))
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=got.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(flag.item() == 1)
