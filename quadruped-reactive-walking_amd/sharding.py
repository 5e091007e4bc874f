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
