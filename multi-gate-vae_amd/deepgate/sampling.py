"""Negative edge sampling for the reconstruction loss.

The reference calls `torch_geometric.utils.negative_sampling(pos ∪ self-loops, N)`
(dg_ae_model_aig.py:115-119): as many (src, dst) pairs as that edge set has, drawn uniformly from the
pairs that are neither an existing edge nor a self loop (pairs may cross graphs of a batch).  Same
distribution here, from rejection sampling on the device with torch index ops (the pairs feed the
HIP recon-loss kernel), or — when the batch's GraphPlan is at hand — from the fused device sampler
(`negative_sampling_device`, csrc/neg_sample.hip), which also buckets the pairs by source and by destination so
that the loss gradient needs no atomics."""
import itertools

import torch

from . import _hip
from ._hip import ptr

_CALLS = itertools.count()


class NegativeEdges:
    """Sampled negative pairs grouped by source: `edge_index` [2, E] int64 plus the two int32 CSRs over them."""

    def __init__(self, edge_index, out_ptr, out_dst, in_ptr, in_src):
        self.edge_index, self.out_ptr, self.out_dst, self.in_ptr, self.in_src = edge_index, out_ptr, out_dst, in_ptr, in_src

    @property
    def csr(self):
        return self.out_ptr, self.out_dst, self.in_ptr, self.in_src


def bucket_negatives(neg, N):
    """NegativeEdges over the pairs `neg` [2, E] (int64, device): both CSRs from the batch builder's CSR kernels (histogram,
    scan, cursor fill, per-list sort by pair id: every list comes out in pair order whatever order the atomics ran in), the pair
    list re-emitted in by-source order.  Deterministic, unlike a plain atomic-cursor bucketing."""
    dev = neg.device
    E = int(neg.shape[1])
    i32 = dict(dtype=torch.int32, device=dev)
    src, dst = neg[0].contiguous(), neg[1].contiguous()
    in_ptr, out_ptr = torch.empty(N + 1, **i32), torch.empty(N + 1, **i32)
    in_src, in_dst, out_dst, out_slot = (torch.empty(max(E, 1), **i32) for _ in range(4))
    n_s = _hip.call_value('mgv_plan_csr_scratch_ints', N, E)
    scratch = torch.empty(n_s, **i32)
    status = torch.empty(2, **i32)
    _hip.call('mgv_plan_csr', N, E, ptr(src), ptr(dst), ptr(in_ptr), ptr(in_src), ptr(in_dst), ptr(out_ptr), ptr(out_dst), ptr(out_slot),
              None, None, ptr(scratch), n_s, ptr(status))
    if int(status[0].item()) != 0:           # caller-supplied pairs, bucketed once and cached: one host read
        raise ValueError('neg_edge_index holds node ids outside [0, num_nodes)')
    counts = (out_ptr[1:] - out_ptr[:-1]).long()
    srt_src = torch.repeat_interleave(torch.arange(N, device=dev), counts, output_size=E)
    srt = torch.stack([srt_src, out_dst[:E].long()])
    return NegativeEdges(srt, out_ptr, out_dst, in_ptr, in_src)


def negative_sampling_device(plan, num_neg_samples=None, generator=None):
    """Same distribution as `negative_sampling` (uniform over non-edges that are not self loops, with replacement),
    drawn by one kernel from a counter-based generator seeded from torch's seed and a call counter (no host sync)."""
    N, dev = plan.N, plan.device
    if getattr(plan, 'num_self_loops', None) is None:
        plan.count_self_loops()              # (one host read-back per plan; GraphPlan.warm does it on the prefetcher's stream for fresh batches)
    E = (plan.E - plan.num_self_loops) + N if num_neg_samples is None else int(num_neg_samples)
    base = generator.initial_seed() if generator is not None else torch.initial_seed()
    seed = (base * 0x9E3779B97F4A7C15 + next(_CALLS) * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D) & 0xFFFFFFFFFFFFFFFF
    scratch = torch.zeros(2, N, dtype=torch.int32, device=dev)          # counts by source / destination
    neg = torch.empty(2, E, dtype=torch.int64, device=dev)
    rank = torch.empty(2, max(E, 1), dtype=torch.int32, device=dev)     # every pair's place inside its two buckets
    _hip.call('mgv_neg_sample', N, E, seed, ptr(plan.out_ptr), ptr(plan.out_dst), ptr(neg[0]), ptr(neg[1]), ptr(scratch[0]), ptr(scratch[1]),
              ptr(rank[0]), ptr(rank[1]))
    ptrs = torch.zeros(2, N + 1, dtype=torch.int32, device=dev)
    for k in range(2):                 # 1-D scans (the device-wide scan; the batched innermost-dim kernel is ~100x slower here)
        ptrs[k, 1:] = torch.cumsum(scratch[k], 0, dtype=torch.int32)
    srt = torch.empty(2, E, dtype=torch.int64, device=dev)
    out_dst = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    in_src = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    _hip.call('mgv_neg_bucket', E, ptr(neg[0]), ptr(neg[1]), ptr(ptrs[0]), ptr(ptrs[1]), ptr(rank[0]), ptr(rank[1]),
              ptr(srt[0]), ptr(srt[1]), ptr(out_dst), ptr(in_src))
    # the buckets were filled in thread-arrival order: sort every list by neighbour id so that the order (and with it every
    # floating-point sum over a list) no longer depends on thread arrival; equal ids are duplicates of one pair
    ss = torch.empty(N + 1 + E, dtype=torch.int32, device=dev)
    _hip.call('mgv_sort_lists_i32', N, E, ptr(ptrs[0]), ptr(out_dst), ptr(ss), ss.numel())
    _hip.call('mgv_sort_lists_i32', N, E, ptr(ptrs[1]), ptr(in_src), ptr(ss), ss.numel())
    srt[1].copy_(out_dst[:E])          # the pair list in by-source order (sources are constant inside a list)
    return NegativeEdges(srt, ptrs[0], out_dst, ptrs[1], in_src)


def sorted_edge_keys(pos_edge_index, num_nodes):
    """Sorted src*N+dst keys of the non-self-loop edges (static per batch: cache them)."""
    src, dst = pos_edge_index[0].long(), pos_edge_index[1].long()
    keep = src != dst
    return torch.sort(src[keep] * int(num_nodes) + dst[keep]).values


def negative_sampling(pos_edge_index, num_nodes, num_neg_samples=None, generator=None, keys=None):
    dev = pos_edge_index.device
    N = int(num_nodes)
    if keys is None:
        keys = sorted_edge_keys(pos_edge_index, N)
    want = int(keys.numel()) + N if num_neg_samples is None else int(num_neg_samples)
    out = torch.empty((2, 0), dtype=torch.long, device=dev)
    guard = 0
    while out.shape[1] < want:
        m = want - out.shape[1]
        m = m + m // 8 + 16
        s = torch.randint(0, N, (m,), device=dev, generator=generator)
        d = torch.randint(0, N, (m,), device=dev, generator=generator)
        k = s * N + d
        ok = s != d
        if keys.numel() > 0:
            pos = torch.searchsorted(keys, k).clamp_(max=keys.numel() - 1)
            ok &= keys[pos] != k
        out = torch.cat([out, torch.stack([s[ok], d[ok]])], dim=1)
        guard += 1
        if guard > 64:
            raise RuntimeError('negative_sampling: graph too dense to draw %d non-edges' % want)
    return out[:, :want].contiguous()
