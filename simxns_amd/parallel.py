"""Data-parallel pieces of the path: one process per GPU, torch.distributed over RCCL ("nccl" on ROCm).

  * gather_with_local_grad -- the cross-GPU in-batch-negative exchange: ONE all-gather of the device-resident
    [CLS] embeddings per tensor, rank order, the local slot keeps its autograd edge and remote rows are
    constants.  Replaces the reference's D2H copy -> pickle -> 5.12 GB uint8 all_reduce -> unpickle -> H2D
    (SimANS/utils/dpr_utils.py:166-228 as called from PROD/ProD_base/train_DE_model_marco.py:224-278).
  * inbatch_nll_allgather  -- caculate_cont_loss on top of it (positives at r*B*(1+N) + j*(1+N)).
  * allreduce_flat_grads   -- DDP's gradient averaging as one all-reduce per tower over the flat f32 buffer
    (the division by W is folded into the optimiser kernel's grad_scale).
  * all_gather_list        -- drop-in for dpr_utils.all_gather_list (arbitrary picklable data, rank order).
xGMI is point-to-point (7 links/GPU): payloads here are 6.7 MB/rank (embeddings, latency-bound) and 438-876 MB
(gradients, ring/tree per-link bound) -- few large collectives, no per-parameter buckets.
"""
import os

import torch
import torch.distributed as dist

# SIMX_FORCE_COLLECTIVES=1: issue the collectives for a one-rank group too (tests/test_dp_gpu.py drives the whole
# data-parallel path through RCCL on a single GPU this way); a production run never sets it
FORCE_COLLECTIVES = os.environ.get("SIMX_FORCE_COLLECTIVES") == "1"


def _world(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


class _GatherLocalGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        W, r = _world(group)
        ctx.rows, ctx.rank = x.shape[0], r
        x = x.contiguous()
        out = x.new_empty((W * x.shape[0],) + tuple(x.shape[1:]))
        dist.all_gather_into_tensor(out, x, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        # remote rows are constants on every rank (train_DE_model_marco.py:251-264): no reduce-scatter
        return g[ctx.rank * ctx.rows:(ctx.rank + 1) * ctx.rows].contiguous(), None


def gather_with_local_grad(x, group=None):
    """[n,...] per rank (equal n on all ranks) -> [W*n,...] in rank order; gradient flows to the local rows only."""
    W, _ = _world(group)
    if W == 1 and not (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized()):
        return x
    return _GatherLocalGrad.apply(x, group)


def global_positive_indices(world_size, batch_per_rank, ctx_per_query):
    """positive_idx_per_question of the rank-ordered concatenation: r*B*D + j*D."""
    D = ctx_per_query
    return [r * batch_per_rank * D + j * D for r in range(world_size) for j in range(batch_per_rank)]


def inbatch_nll_allgather(q, ctx_vectors, ctx_per_query, loss_scale=None, group=None):
    """BiEncoderNllLoss over the global [W*B, W*B*D] score matrix with the reference's gather semantics."""
    from . import ops
    W, r = _world(group)
    B = q.shape[0]
    gq, gc = gather_with_local_grad(q, group), gather_with_local_grad(ctx_vectors, group)
    pos = global_positive_indices(W, B, ctx_per_query)
    loss, _ = ops.inbatch_nll_loss(gq, gc, pos, loss_scale, local_q=(r * B, B), local_ctx=(r * B * ctx_per_query, B * ctx_per_query))
    return loss


def allreduce_flat_grads(buffers, group=None, async_op=False):
    """SUM all-reduce of every flat gradient buffer; returns (handles, grad_scale = 1/W)."""
    W, _ = _world(group)
    if W == 1:
        return [], 1.0
    handles = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group, async_op=async_op) for b in buffers]
    return handles, 1.0 / W


def all_gather_list(data, group=None, max_size=16384):
    """Drop-in for SimANS/utils/dpr_utils.py:166-228: gathers arbitrary picklable data from all ranks into a
    list (rank order).  ``max_size`` is accepted for signature compatibility; no fixed byte buffer is used."""
    W, _ = _world(group)
    if W == 1:
        return [data]
    out = [None] * W
    dist.all_gather_object(out, data, group=group)
    return out
