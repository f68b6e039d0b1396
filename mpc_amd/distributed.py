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

from .binding import winner_record_words

I64_MAX = np.iinfo(np.int64).max


def shard_range(n_total, rank, world):
    """Contiguous ordinal range [first, first + count) of `rank`: sizes differ by at most one."""
    base, extra = divmod(int(n_total), int(world))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def _global_ordinal(local, offset):
    """local winning ordinals -> global ones; -1 (UINT64_MAX: nothing selected, no finite score) -> INT64_MAX"""
    return torch.where(local == -1, torch.full_like(local, I64_MAX), local + offset)


def combine_winners(winners, ordinal_offset=0, group=None, always=False):
    """winners: int64 tensor [P, 2] (or [2]) on this rank = (score bits, local winning ordinal) per problem, as
    mpc_hip_copy_winners leaves it.  ordinal_offset: what to add to the local ordinals to make them global (0 if
    the library was given first_ordinal).  Returns (score_bits [P], global_ordinal [P]), identical on all ranks.
    Asynchronous with respect to the host on CUDA tensors (ordered on the current stream).  always=True runs the
    collectives even in a one-rank group (rehearsal)."""
    w = winners.view(-1, 2)
    key = w[:, 0].clone()
    if dist.is_available() and dist.is_initialized() and (always or dist.get_world_size(group) > 1):
        dist.all_reduce(key, op=dist.ReduceOp.MIN, group=group)
        cand = torch.where(w[:, 0] == key, _global_ordinal(w[:, 1], ordinal_offset), torch.full_like(key, I64_MAX))
        dist.all_reduce(cand, op=dist.ReduceOp.MIN, group=group)
    else:
        cand = _global_ordinal(w[:, 1], ordinal_offset)
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


def combine_gathered(gathered, n_shards, n_problems, ordinal_stride=0):
    """torch form of mpc_hip_combine_shards (used on CPU tensors by the gloo tests; on the GPU the library's kernel
    does this in one launch): gathered int64 [n_shards, n_problems, 2] -> (score_bits [P], global_ordinal [P])."""
    g = gathered.view(n_shards, n_problems, 2)
    keys = g[:, :, 0]
    kmin = keys.min(dim=0).values
    off = (torch.arange(n_shards, dtype=torch.int64, device=g.device) * ordinal_stride).view(-1, 1)
    ords = torch.where(g[:, :, 1] == -1, torch.full_like(keys, I64_MAX), g[:, :, 1] + off)
    cand = torch.where(keys == kmin, ords, torch.full_like(keys, I64_MAX)).min(dim=0).values
    return kmin, cand


class ShardCombiner:
    """The combine step for throughput runs of independent solves (bench.py --gpus N), batched and pipelined:

        combiner.begin() ; solver.run_*(...) ; combiner.submit()        # per solve
        key_bits, ordinal = combiner.flush()         # after the last one: every submitted solve is combined

    begin() points the solver's winner sink at the next row of a batch buffer, so the finish kernel itself deposits
    this rank's winner records there (no device copy per solve); every `batch` solves ONE asynchronous all-gather moves
    the batch (instead of two dependent 8-byte all-reduces per solve: fewer, larger collectives), and its latency hides
    behind the next batch of rollouts; then one small kernel decides:
        exact=True   the records carry the winners' controls (mpc_hip_set_winner_record_sink, 16 + 16 * depth bytes
                     per problem) and mpc_hip_combine_shards_exact re-scores, with the strict model, the winners of
                     shards within MPC_REFINE_REL of a row's minimum: the reference's first strict minimum across the
                     GPUs, from ONE exchange
        exact=False  16-byte records, mpc_hip_combine_shards on the fast keys (two shards whose winners lie within the
                     fast model's ~1e-7 error may then resolve differently from the reference)
    No host synchronisation anywhere over RCCL.  Over a backend that moves host memory only (gloo: the CPU tests, and the
    rehearsal of N ranks on ONE GPU, which RCCL refuses) the batch is staged through pinned host buffers -- device ->
    host copy on the stream, ONE stream synchronisation per batch, the gather, host -> device copy before the combine
    kernel -- the solver must then run on torch's current stream.  ordinal_stride: what separates the shards' local ordinals (0 when the shards
    were run with their first_ordinal).  depth: levels per record, >= the deepest resident problem.  The exact decision
    reads the resident problems: flush() before they change.
    flush() returns the combined (score bits, global ordinal) of the LAST solve; `results` holds the last batch.
    """

    def __init__(self, solver, n_problems, ordinal_stride=0, group=None, batch=16, slots=3, always=False, exact=True,
                 depth=None, precision=0, selection=0):
        self.solver, self.P, self.stride, self.group = solver, int(n_problems), int(ordinal_stride), group
        self.precision, self.selection = int(precision), int(selection)     # of the runs whose records are combined
        self.B = int(batch)
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.collective = on and (always or self.world > 1)
        self.exact = bool(exact) and self.collective
        if self.exact and depth is None:
            raise ValueError("ShardCombiner(exact=True) needs depth (levels per winner record)")
        self.depth = int(depth) if self.exact else 0
        self.W = winner_record_words(self.depth) if self.exact else 2
        dev = torch.device("cuda", solver.device)
        row = self.W * self.P
        z = lambda n: torch.zeros(n, dtype=torch.int64, device=dev)
        self.win = [z(self.B * row) for _ in range(slots)]
        for w in self.win:                   # rows never written read "nothing selected" (+inf bits, ordinal -1), not score 0
            w.view(-1, self.W)[:, 0] = 0x7FF0000000000000
            w.view(-1, self.W)[:, 1] = -1
        torch.cuda.synchronize(dev)          # the solver's stream may not be torch's
        self.gathered = [z(self.B * row * self.world) if self.collective else None for _ in range(slots)]
        # a backend without device collectives (gloo): the exchange goes through pinned host memory
        self.host_staged = self.collective and dist.get_backend(group) != "nccl"
        if self.host_staged:
            hz = lambda n: torch.zeros(n, dtype=torch.int64).pin_memory()
            self.win_h = [hz(self.B * row) for _ in range(slots)]
            self.gathered_h = [hz(self.B * row * self.world) for _ in range(slots)]
        self.results = z(self.B * 2 * self.P)
        self.work = [None] * slots
        self.pending = []                    # slots whose batch is not combined yet, oldest first
        self.slot, self.fill, self.last = 0, 0, None

    def _combine(self, r):
        rows = self.B * self.P
        if self.work[r] is not None:
            self.work[r].wait()              # the current stream waits for the collective (no host sync)
            self.work[r] = None
            if self.host_staged:
                self.gathered[r].copy_(self.gathered_h[r], non_blocking=True)
        if self.exact:
            self.solver.combine_shards_exact(self.gathered[r], self.world, rows, self.depth, self.stride, self.results,
                                             self.precision, self.selection)
        else:
            src = self.gathered[r] if self.collective else self.win[r]
            self.solver.combine_shards(src, self.world if self.collective else 1, rows, self.stride, self.results)
        self.pending.remove(r)

    def tied_rows(self):
        """rows (solve x problem) that had a cross-shard near-tie so far; synchronises"""
        return self.solver.tied_rows() if self.exact else 0

    def _close_batch(self):
        r = self.slot
        self.solver.flush()                  # a deferred finish (mpc_hip_set_pipeline) deposits the batch's last row now
        if self.fill < self.B:               # a partial batch: the rows behind it still hold an OLDER batch's records in a
            tail = self.win[r].view(self.B * self.P, self.W)[self.fill * self.P:]      # reused slot -- "nothing selected" again,
            tail[:, 0] = 0x7FF0000000000000                                            # so that they are neither re-scored nor
            tail[:, 1] = -1                                                            # counted as near-tie rows
        if self.host_staged:
            self.win_h[r].copy_(self.win[r], non_blocking=True)
            torch.cuda.current_stream(self.win[r].device).synchronize()    # the host collective reads the pinned copy
            self.work[r] = dist.all_gather_into_tensor(self.gathered_h[r], self.win_h[r], group=self.group, async_op=True)
        elif self.collective:
            self.work[r] = dist.all_gather_into_tensor(self.gathered[r], self.win[r], group=self.group, async_op=True)
        self.pending.append(r)
        self.last = (r, self.fill - 1)
        self.slot, self.fill = (r + 1) % len(self.win), 0

    def _row(self):
        if self.fill == 0 and self.slot in self.pending:      # about to overwrite a batch that is not combined yet
            while self.slot in self.pending:
                self._combine(self.pending[0])
        row = self.W * self.P
        return self.win[self.slot][self.fill * row:(self.fill + 1) * row]

    def _point_sink(self, dst):
        if self.exact:
            self.solver.set_winner_record_sink(dst, self.depth)
        else:
            self.solver.set_winner_sink(dst)

    def begin(self):
        """before the run whose winners go into the next row"""
        self._point_sink(self._row())
        self._begun = True

    def submit(self):
        """after the run"""
        if not getattr(self, "_begun", False):        # begin() was skipped: a device copy of the winners instead
            if self.exact:
                self.solver.copy_winner_records(self._row(), self.depth)
            else:
                self.solver.copy_winners(self._row())
        self._begun = False
        self.fill += 1
        if self.fill == self.B:
            older = list(self.pending)
            self._close_batch()
            for r in older:                           # the batches before the one just closed: their gather has had a batch of rollouts
                self._combine(r)

    def flush(self):
        self._point_sink(None)
        if self.fill:
            self._close_batch()
        while self.pending:
            self._combine(self.pending[0])
        if self.last is None:
            return None
        o = self.results.view(self.B, self.P, 2)[self.last[1]]
        return o[:, 0], o[:, 1]


def combine_shards_exact(solver, records, depth, ordinal_stride=0, precision=0, selection=0):
    """The exact cross-shard decision with every shard in THIS process (rehearsal of N GPUs on one): records int64
    [n_shards, rows, winner_record_words(depth)] on the device (copy_winner_records of each shard's run, shard-local
    ordinals as the runs reported them; torch.stack in place of the all-gather) -> int64 [rows, 2] = (strict-decided
    score bits, global ordinal).  solver: any handle holding the problems (every rank holds them all)."""
    n_shards, rows = int(records.shape[0]), int(records.shape[1])
    g = records.contiguous()
    torch.cuda.synchronize()
    out = torch.zeros((rows, 2), dtype=torch.int64, device=g.device)
    solver.combine_shards_exact(g, n_shards, rows, depth, ordinal_stride, out, precision, selection)
    solver.synchronize()
    return out


def combine_winners_exact_device(solver, depth, ordinal_stride=0, group=None, precision=0, selection=0):
    """One solve, every rank its own GPU: the winner records of the solver's last run -> ONE all-gather ->
    mpc_hip_combine_shards_exact, all stream-ordered on the device (the solver must run on torch's current stream).
    -> int64 [P, 2] (score bits, global ordinal), identical on every rank."""
    P, W = solver._n_problems, winner_record_words(depth)
    dev = torch.device("cuda", solver.device)
    world = dist.get_world_size(group)
    mine = torch.empty(P * W, dtype=torch.int64, device=dev)
    solver.copy_winner_records(mine, depth)
    gathered = torch.empty(world * P * W, dtype=torch.int64, device=dev)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(gathered, mine, group=group)
    else:                                    # host collectives only (gloo): through host memory, one synchronisation
        mine_h = mine.cpu()
        gathered_h = torch.empty(world * P * W, dtype=torch.int64)
        dist.all_gather_into_tensor(gathered_h, mine_h, group=group)
        gathered.copy_(gathered_h)
    out = torch.empty((P, 2), dtype=torch.int64, device=dev)
    solver.combine_shards_exact(gathered, world, P, depth, ordinal_stride, out, precision, selection)
    return out


def combine_winners_exact(winners, ordinal_offset, strict_score, group=None, refine_rel=1.0e-6, floor=0.0):
    """The fast mode's cross-shard decision made exact.  Inside a shard, candidates within MPC_REFINE_REL of the
    shard's minimum are already decided by the strict model; ACROSS shards the packed winners carry fast scores, which
    are only good to ~1e-7.  Here the winners are all-gathered; wherever the best two shards of a problem lie within
    refine_rel of each other, each of those shards re-scores ITS winner with the strict model -- strict_score(problem
    index, local ordinal) -> float, e.g. a one-candidate mpc_hip_eval_* call in MPC_PREC_F64 -- and the strict (score,
    global ordinal) decides, as the reference's first strict minimum would (controller.cpp:158-166).  Costs one host
    synchronisation, and two more 8-byte collectives only when such a tie exists.  -> (score bits [P], global
    ordinal [P]) as int64 tensors on the winners' device, identical on every rank."""
    w = winners.view(-1, 2)
    P = w.shape[0]
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if not on:
        return w[:, 0].clone(), _global_ordinal(w[:, 1], ordinal_offset)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = torch.stack([w[:, 0], _global_ordinal(w[:, 1], ordinal_offset)], dim=1).contiguous()
    gathered = torch.empty((world, P, 2), dtype=torch.int64, device=w.device)
    dist.all_gather_into_tensor(gathered.view(-1), mine.view(-1), group=group)
    g = gathered.cpu().numpy()
    keys = g[:, :, 0].view(np.float64)                              # non-negative scores
    kmin = keys.min(axis=0)
    # the window of mpc_refine_limit (mpc_internal.h); floor = 1 % of the problem's weights' sum (scalar or [P])
    with np.errstate(invalid="ignore"):
        near = keys <= kmin + np.maximum(kmin, floor) * refine_rel  # [world, P]; +inf rows stay out unless all are
    near &= np.isfinite(keys)
    tied = np.flatnonzero(near.sum(axis=0) > 1)
    strict = torch.full((P,), I64_MAX, dtype=torch.int64, device=w.device)
    if tied.size:                                                   # the same decision on every rank
        vals = np.full(P, np.inf)
        for p in tied:
            if near[rank, p]:
                vals[p] = float(strict_score(int(p), int(w[p, 1].item())))
        strict = torch.from_numpy(np.where(np.isfinite(vals), vals, np.inf).view(np.int64).copy()).to(w.device)
        skey = strict.clone()
        dist.all_reduce(skey, op=dist.ReduceOp.MIN, group=group)
        cand = torch.where(strict == skey, mine[:, 1], torch.full_like(skey, I64_MAX))
        dist.all_reduce(cand, op=dist.ReduceOp.MIN, group=group)
        skey, cand = skey.cpu().numpy(), cand.cpu().numpy()
    out_k = np.empty(P, dtype=np.int64); out_o = np.empty(P, dtype=np.int64)
    for p in range(P):
        if tied.size and p in tied:
            out_k[p], out_o[p] = skey[p], cand[p]
        else:
            col = [(int(g[s, p, 0]), int(g[s, p, 1])) for s in range(world)]
            out_k[p], out_o[p] = min(col)
    return torch.from_numpy(out_k).to(w.device), torch.from_numpy(out_o).to(w.device)


# ---- tree search split over GPUs (mpc_hip_solve_tree_shard) ------------------------------------------------------
def tree_levels_tensor(levels, n_levels, device="cpu"):
    """this shard's per-level record [n_levels, 3] int64: leaves, winning score bits, winning LOCAL leaf ordinal"""
    rows = []
    for l in range(n_levels):
        bits = int(np.float64(levels[l].best_score).view(np.int64))
        rows.append([int(levels[l].n_leaves), bits, int(np.uint64(levels[l].best_leaf).view(np.int64))])
    return torch.tensor(rows, dtype=torch.int64, device=device)


def combine_tree_levels(gathered, split_level):
    """gathered: int64 [n_shards, n_levels, 3] (tree_levels_tensor of every shard, shard order = slice order).
    -> list per level of (total leaves, global DFS ordinal of the winner, score bits, owning shard).  Levels up to
    split_level were expanded in full by every shard (shard 0's record is the level's); below it a shard's ordinals
    are local to its leaves, and the shards' leaves are consecutive runs of the full level in shard order, so
    global = local + the lower shards' leaves; the winner is the (score, global ordinal) minimum: the reference's
    first strict minimum in DFS order (controller.cpp:158-166)."""
    g = np.asarray(gathered.cpu().numpy() if hasattr(gathered, "cpu") else gathered, dtype=np.int64)
    n_shards, n_levels, _ = g.shape
    out = []
    for l in range(n_levels):
        if l < split_level:
            out.append((int(g[0, l, 0]), int(g[0, l, 2]), int(g[0, l, 1]), 0))
            continue
        prefix = np.concatenate([[0], np.cumsum(g[:, l, 0])[:-1]])
        best = None
        for s in range(n_shards):
            if g[s, l, 0] == 0 or g[s, l, 2] == -1:        # no leaves / no leaf with a finite score in this shard
                continue
            cand = (int(g[s, l, 1]), int(prefix[s] + g[s, l, 2]), s)          # non-negative scores order like their bits
            if best is None or cand[:2] < best[:2]:
                best = cand
        if best is None:                                   # nothing selected at this level anywhere
            best = (int(np.float64(np.inf).view(np.int64)), -1, 0)
        out.append((int(g[:, l, 0].sum()), best[1], best[0], best[2]))
    return out


def solve_tree_sharded(solver, problem, max_levels, split_level, precision=1, group=None, pieces_per_rank=1):
    """Every rank of `group` runs its shard of the search on its GPU, the per-level records are all-gathered (24 bytes
    per level, piece and rank) and combined.  -> (combined levels, this rank's Result, owner rank of the final winner).
    pieces_per_rank > 1: the search is cut into world x pieces_per_rank shards and a rank runs its consecutive
    pieces_per_rank of them one after the other (a level too wide for one expansion's 32-bit ordinals even as a
    world-th: depth 9 of the reference's control sets -- 8.8e10 leaves -- on 8 GPUs x 5 pieces); the Result returned is
    that of the rank's best piece."""
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    ppr = int(pieces_per_rank)
    assert ppr >= 1
    runs = [solver.solve_tree(problem, max_levels, precision, split_level=split_level, shard=rank * ppr + k, n_shards=world * ppr)
            for k in range(ppr)]
    dev = "cuda" if on and dist.get_backend(group) == "nccl" else "cpu"
    n_levels = min(int(r.levels) for r, _ in runs)
    if world > 1:
        # shards expand slices of different widths: a node budget, the storage cap or memory can stop them at different
        # depths.  Only the levels EVERY shard completed can be combined (and the gather needs equal shapes).
        common = torch.tensor([n_levels], dtype=torch.int64, device=dev)
        dist.all_reduce(common, op=dist.ReduceOp.MIN, group=group)
        n_levels = int(common.item())
    mine = torch.stack([tree_levels_tensor(lv, n_levels, dev).view(n_levels, 3) for _, lv in runs])      # [ppr, n_levels, 3]
    if world > 1:
        gathered = torch.empty((world,) + tuple(mine.shape), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(gathered.view(-1), mine.contiguous().view(-1), group=group)
        gathered = gathered.view(world * ppr, n_levels, 3)                                                 # shard order = slice order
    else:
        gathered = mine
    combined = combine_tree_levels(gathered, split_level)
    owner_shard = combined[-1][3]
    # this rank's own best piece (by the last common level's score, then ordinal: pieces are in DFS order)
    last = n_levels - 1
    best = min(range(ppr), key=lambda k: (int(mine[k, last, 1]) if int(mine[k, last, 2]) != -1 else 2 ** 63 - 1, k))
    res = runs[owner_shard - rank * ppr][0] if rank * ppr <= owner_shard < (rank + 1) * ppr else runs[best][0]
    return combined, res, owner_shard // ppr
