"""Multi-GPU combine step of the hot path: one process per GPU, candidates sharded by
contiguous ordinal range, shards combined by a min-allreduce of the packed winner over torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The exchange is 8 bytes per problem and rank per collective -- pure latency, so it is two tiny all-reduces and
never a ring over a large buffer.  Scores are non-negative doubles (controller.cpp:240 with weights >= 0), so
their bit patterns order like the values and int64 MIN is the argmin; the second all-reduce takes the lowest
GLOBAL ordinal among the ranks that hold the minimum, which is the reference's first-strict-minimum rule
(controller.cpp:158-166) because ordinals grow with rank.
"""
import numpy as np
import torch
import torch.distributed as dist

I64_MAX = np.iinfo(np.int64).max


def shard_range(n_total, rank, world):
    """Contiguous ordinal range [first, first + count) of `rank`: sizes differ by at most one."""
    base, extra = divmod(int(n_total), int(world))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def combine_winners(winners, ordinal_offset=0, group=None):
    """winners: int64 tensor [P, 2] (or [2]) on this rank = (score bits, local winning ordinal) per problem, as
    mpc_hip_copy_winners leaves it.  ordinal_offset: what to add to the local ordinals to make them global (0 if
    the library was given first_ordinal).  Returns (score_bits [P], global_ordinal [P]), identical on all ranks.
    Asynchronous with respect to the host on CUDA tensors (ordered on the current stream)."""
    w = winners.view(-1, 2)
    key = w[:, 0].clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(key, op=dist.ReduceOp.MIN, group=group)
        cand = torch.where(w[:, 0] == key, w[:, 1] + ordinal_offset, torch.full_like(key, I64_MAX))
        dist.all_reduce(cand, op=dist.ReduceOp.MIN, group=group)
    else:
        cand = w[:, 1] + ordinal_offset
    return key, cand


def owner_rank(global_ordinal, n_total, world):
    """Rank whose shard (shard_range) contains global_ordinal."""
    base, extra = divmod(int(n_total), int(world))
    g = int(global_ordinal)
    split = extra * (base + 1)
    if g < split:
        return g // (base + 1)
    return extra + (g - split) // base if base else world - 1


def bits_to_score(bits):
    return np.asarray(bits.cpu().numpy() if hasattr(bits, "cpu") else bits, dtype=np.int64).view(np.float64)
