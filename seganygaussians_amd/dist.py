"""View-sharded multi-GPU step (BASELINE config 4).

The rasterizer hot path shards over VIEWS: every rank holds the full (replicated) Gaussian parameters,
renders and back-propagates its own camera(s), and the only exchange is the SUM of the per-Gaussian
gradients before the optimizer step.  One process per GPU, `torch.distributed` ("nccl" == RCCL over xGMI on
ROCm; "gloo" on CPU for tests).  The reference itself has no distributed layer (single process, cuda:0);
this is the new batch semantic "gradient = sum over the N views of an iteration".

What is exchanged: in SAGA's contrastive training only `_point_features` is optimised
(scene/gaussian_model_ff.py:154-162), so the message is dL/d(features): P x C fp32 (128 MB at 1M x 32).
`allreduce_grads` sends each tensor as ONE flat bucket -- on MI355X's point-to-point xGMI mesh large
messages let RCCL drive all 7 links (ring algorithms are per-link bound), so a few big collectives beat many
small ones.  Passing several tensors coalesces them into a single flat buffer first.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence  # noqa: F401

import torch
import torch.distributed as dist


def views_for_rank(num_views: int, rank: int, world_size: int) -> List[int]:
    """Round-robin view ownership: rank r renders views r, r + world, r + 2*world, ...."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, num_views, world_size))


def allreduce_grads(tensors: Sequence[Optional[torch.Tensor]], group=None, average: bool = False) -> None:
    """In-place SUM (or mean) all-reduce of gradient tensors across ranks, coalesced into one flat bucket per
    dtype/device.  `None` entries are skipped.  No-op when torch.distributed is not initialised (1 GPU)."""
    ts = [t for t in tensors if t is not None]
    if not ts or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    if len(ts) == 1 and ts[0].is_contiguous():
        dist.all_reduce(ts[0], op=dist.ReduceOp.SUM, group=group)
        if average:
            ts[0].div_(world)
        return
    buckets = {}
    for t in ts:
        buckets.setdefault((t.dtype, t.device), []).append(t)
    for (_dt, _dev), group_ts in buckets.items():
        flat = torch.cat([t.reshape(-1) for t in group_ts])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(world)
        off = 0
        for t in group_ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


_side_streams = {}


def allreduce_grads_async(tensors: Sequence[Optional[torch.Tensor]], group=None):
    """Starts the in-place SUM all-reduce of contiguous gradient tensors WITHOUT making the current stream wait for it.
    Returns (event, keepalive): `event` is a torch.cuda.Event that fires when the reduced values are in place (None on
    CPU tensors, where the call completes the reduction, and when there is nothing to reduce); `keepalive` must be held
    until the event has been waited for.  Use with rasterizer.set_features_ready_event: in SAGA's feature training the
    next view's geometry stages (preprocess, binning, per-tile sort: a fifth of a step) do not depend on
    the reduced gradients and can run while they travel."""
    ts = [t for t in tensors if t is not None]
    if not ts or not (dist.is_available() and dist.is_initialized()):
        return None, None
    works = []
    for t in ts:
        if not t.is_contiguous():
            raise ValueError("allreduce_grads_async needs contiguous tensors")
        works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True))
    if not ts[0].is_cuda:
        for w in works:
            w.wait()
        return None, None
    dev = ts[0].device
    side = _side_streams.get(dev)
    if side is None:
        side = _side_streams[dev] = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        for w in works:
            w.wait()  # NCCL/RCCL: the SIDE stream waits for the collective, the host and the compute stream do not
        ev = torch.cuda.Event()
        ev.record(side)
    return ev, (ts, works)


def shard_range(numel: int, rank: int, world: int):
    """Elements [lo, hi) of a flat tensor of `numel` elements that rank `rank` owns: equal shards of ceil(numel / world)
    elements (the last ones shorter or empty) -- the layout reduce_scatter_tensor / all_gather_into_tensor use on a tensor
    padded to world * ceil(numel / world)."""
    per = -(-numel // world)
    return min(numel, rank * per), min(numel, (rank + 1) * per)


def sharded_update_async(param: torch.Tensor, grad: torch.Tensor, update_shard, group=None, average: bool = False):
    """The exchange of a view-sharded step as reduce-scatter -> rank-local update -> all-gather instead of an all-reduce followed
    by the same update on every rank:

        all-reduce :  every rank receives all P x C summed gradients (2 (N-1)/N S bytes per rank on a ring), then every rank runs
                      the optimizer over all P rows;
        this       :  rank r receives the SUM of its 1/N of the rows ((N-1)/N S), runs `update_shard(param_rows, grad_rows, lo, hi)`
                      on them alone (optimizer state stays sharded: 1/N of Adam's moments per rank) and the UPDATED rows are
                      gathered ((N-1)/N S).

    The bytes on the wire are the same; what changes is what has to finish before the next view's blend stage: only the all-gather
    of the updated rows -- the reduce-scatter starts as soon as this view's feature gradients are complete and runs beside the
    geometry backward, and the optimizer touches 1/N of the rows.  `param` / `grad`: contiguous tensors of the same shape
    (SAGA: `_point_features` and its .grad, P x C fp32); `update_shard` gets flat views of this rank's elements [lo, hi) of both
    (grad already summed over the ranks) and updates the parameter view IN PLACE.  On return `param` holds the updated values of
    every rank's shard once the returned event has fired (CUDA; the call completes everything on CPU tensors and returns
    (None, None)).  Hand the event to rasterizer.set_features_ready_event like allreduce_grads_async's.  Without an initialised
    process group the update runs over the whole tensor.  `average`: the summed gradient is divided by the world size first
    (allreduce_grads / ViewShardedStep have the same switch).

    STREAMS -- the caller's part of the contract.  Until the returned event has fired, a side stream reads `grad` and WRITES `param`
    in place (record_stream only keeps the allocator from reusing their memory):
      * do not touch `grad` in place before then -- drop it (`param.grad = None`, what optimizer.zero_grad() does by default), never
        `zero_grad(set_to_none=False)` or an in-place op on it;
      * every other reader of `param` on the compute stream -- a feature regulariser, evaluation, save_ply -- must first make that
        stream wait: `torch.cuda.current_stream().wait_event(ev)`.  The next rasterizer forward does this itself when the event was
        handed to rasterizer.set_features_ready_event (it waits right before its blend stage)."""
    if not (param.is_contiguous() and grad.is_contiguous()) or param.shape != grad.shape:
        raise ValueError("sharded_update_async needs contiguous param / grad of one shape")
    n = param.numel()
    pf, gf = param.detach().view(-1), grad.detach().view(-1)
    if not (dist.is_available() and dist.is_initialized()):
        update_shard(pf, gf, 0, n)
        return None, None
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if "nccl" in str(dist.get_backend(group)) and not param.is_cuda:
        raise ValueError("sharded_update_async: an RCCL-only group cannot exchange host tensors")
    per = -(-n // world)
    lo, hi = shard_range(n, rank, world)
    # RCCL (also in a mixed "cuda:nccl,cpu:gloo" group), or gloo on host tensors
    native = (param.is_cuda and "nccl" in str(dist.get_backend(group))) or not param.is_cuda

    def run():
        if per * world == n:
            gin, pall = gf, pf
        else:   # pad to equal shards (never on the benchmark sizes: P x C is a multiple of 8)
            gin = torch.zeros(per * world, dtype=gf.dtype, device=gf.device)
            gin[:n].copy_(gf)
            pall = torch.empty(per * world, dtype=pf.dtype, device=pf.device)
            pall[:n].copy_(pf)
        pshard = pall[rank * per:(rank + 1) * per]
        if native:
            gshard = torch.empty(per, dtype=gf.dtype, device=gf.device)
            dist.reduce_scatter_tensor(gshard, gin, op=dist.ReduceOp.SUM, group=group)
            if average:
                gshard.div_(world)
            update_shard(pshard[:hi - lo], gshard[:hi - lo], lo, hi)
            dist.all_gather_into_tensor(pall, pshard, group=group)   # in place: this rank's shard is where it belongs already
        else:
            # gloo on device tensors (the tests' two ranks on one GPU) has neither collective: the same exchange out of the ones
            # it has -- every rank still updates only its own rows and receives the others' updated rows
            gsum = gin.clone()
            dist.all_reduce(gsum, op=dist.ReduceOp.SUM, group=group)
            gshard = gsum[rank * per:(rank + 1) * per]
            if average:
                gshard.div_(world)
            update_shard(pshard[:hi - lo], gshard[:hi - lo], lo, hi)
            parts = [torch.empty_like(pshard) for _ in range(world)]
            dist.all_gather(parts, pshard.clone(), group=group)
            for r, part in enumerate(parts):
                pall[r * per:(r + 1) * per].copy_(part)
        if pall is not pf:
            pf.copy_(pall[:n])
        return gshard, gin, pall

    if not param.is_cuda:
        run()
        return None, None
    dev = param.device
    side = _side_streams.get(dev)
    if side is None:
        side = _side_streams[dev] = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))   # the gradients of this view are complete
    with torch.cuda.stream(side):
        keep = run()     # collectives are enqueued behind the side stream's work; neither the host nor the compute stream waits
        ev = torch.cuda.Event()
        ev.record(side)
    for t in (param, grad):
        t.record_stream(side)
    return ev, keep


class ShardedAdam:
    """Adam over this rank's shard of ONE parameter tensor, for `sharded_update_async`: the update torch.optim.Adam(lr, betas, eps)
    applies to the whole tensor (no weight decay, no amsgrad: what SAGA's feature training uses, scene/gaussian_model_ff.py:154-162),
    with both moment buffers allocated for this rank's 1 / world of the ELEMENTS only (shard_range cuts the flat tensor, not rows: the
    bounds move with the world size) -- at 1 M x 32 features and N = 8, 32 MB of optimizer state per rank instead of 256 MB, and an
    eighth of the optimizer's memory traffic per step.  state_dict() / load_state_dict() gather / scatter the moments to and from the
    full tensor's layout -- the entries torch.optim.Adam keeps for that parameter ("step", "exp_avg", "exp_avg_sq") -- so that the
    reference's checkpoints (FeatureGaussianModel.capture / restore, scene/gaussian_model_ff.py:403-437) carry the moments and a run
    may resume under another world size.

        opt = ShardedAdam(lr=0.0025)
        ev, keep = sharded_update_async(features, features.grad, opt)     # features: the replicated parameter
        rasterizer.set_features_ready_event(ev)                          # the next forward's blend stage waits for the all-gather only
    """

    def __init__(self, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.step = 0
        self.exp_avg = None
        self.exp_avg_sq = None
        self.bounds = None

    def __call__(self, prow: torch.Tensor, grow: torch.Tensor, lo: int, hi: int) -> None:
        if self.exp_avg is None:
            self.exp_avg, self.exp_avg_sq, self.bounds = torch.zeros_like(prow), torch.zeros_like(prow), (lo, hi)
        if self.bounds != (lo, hi):
            raise ValueError(f"ShardedAdam was created for rows {self.bounds}, called with {(lo, hi)}: one instance per parameter and group")
        self.step += 1
        b1, b2 = self.betas
        self.exp_avg.mul_(b1).add_(grow, alpha=1.0 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(grow, grow, value=1.0 - b2)
        bias1, bias2 = 1.0 - b1 ** self.step, 1.0 - b2 ** self.step
        denom = (self.exp_avg_sq.sqrt() / (bias2 ** 0.5)).add_(self.eps)
        prow.addcdiv_(self.exp_avg, denom, value=-self.lr / bias1)

    def state_dict(self, numel: int, shape=None, group=None) -> dict:
        """The state torch.optim.Adam holds for the parameter (`numel` elements, viewed as `shape`): {"step", "exp_avg", "exp_avg_sq"}
        with FULL moment tensors, gathered from the ranks' shards -- a collective when a process group is up: call it on every rank."""
        full = []
        for buf in (self.exp_avg, self.exp_avg_sq):
            if not (dist.is_available() and dist.is_initialized()):
                full.append(torch.zeros(numel) if buf is None else buf.detach().clone())
                continue
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            per = -(-numel // world)
            lo, hi = shard_range(numel, rank, world)
            if buf is None:
                raise ValueError("ShardedAdam.state_dict: no step taken yet on this rank")
            if self.bounds != (lo, hi):
                raise ValueError(f"ShardedAdam holds elements {self.bounds}; a tensor of {numel} elements gives this rank {(lo, hi)}")
            mine = torch.zeros(per, dtype=buf.dtype, device=buf.device)
            mine[:hi - lo].copy_(buf)
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine, group=group)
            full.append(torch.cat(parts)[:numel])
        out = {"step": torch.tensor(float(self.step)), "exp_avg": full[0], "exp_avg_sq": full[1]}
        if shape is not None:
            out["exp_avg"], out["exp_avg_sq"] = out["exp_avg"].view(shape), out["exp_avg_sq"].view(shape)
        return out

    def load_state_dict(self, state: dict, group=None) -> None:
        """Takes {"step", "exp_avg", "exp_avg_sq"} with FULL moment tensors (this class's state_dict, or torch.optim.Adam's state of
        the parameter) and keeps this rank's elements of them; any world size."""
        ea, es = state["exp_avg"], state["exp_avg_sq"]
        if ea.shape != es.shape:
            raise ValueError("ShardedAdam.load_state_dict: exp_avg and exp_avg_sq differ in shape")
        numel = ea.numel()
        world, rank = (dist.get_world_size(group), dist.get_rank(group)) if dist.is_available() and dist.is_initialized() else (1, 0)
        lo, hi = shard_range(numel, rank, world)
        if self.bounds is not None and self.bounds != (lo, hi):
            raise ValueError(f"ShardedAdam holds elements {self.bounds}; the loaded state ({numel} elements, world {world}) gives this rank {(lo, hi)}")
        dev = self.exp_avg.device if self.exp_avg is not None else ea.device
        self.exp_avg = ea.detach().reshape(-1)[lo:hi].to(dev).clone()
        self.exp_avg_sq = es.detach().reshape(-1)[lo:hi].to(dev).clone()
        self.bounds = (lo, hi)
        self.step = int(float(state["step"]))


class ViewShardedStep:
    """One training iteration over `num_views` views sharded across the ranks.

    `render_backward(view_index)` must render that view with the drop-in rasterizer and call `.backward()`
    (gradients accumulate in the `.grad` of `params`); this class zeroes the grads, runs the local views,
    and sums the gradients over ranks.  The optimizer step stays with the caller (it is identical on every
    rank because parameters and summed gradients are)."""

    def __init__(self, params: Iterable[torch.Tensor], group=None, average: bool = False):
        self.params = [p for p in params]
        self.group = group
        self.average = average

    @property
    def rank(self) -> int:
        return dist.get_rank(self.group) if dist.is_available() and dist.is_initialized() else 0

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def __call__(self, num_views: int, render_backward) -> List[int]:
        for p in self.params:
            p.grad = None
        mine = views_for_rank(num_views, self.rank, self.world_size)
        for v in mine:
            render_backward(v)
        for p in self.params:
            if p.grad is None:  # a rank without views (num_views < world) still joins the collective
                p.grad = torch.zeros_like(p)
        allreduce_grads([p.grad for p in self.params], group=self.group, average=self.average)
        return mine
