"""Negative edge sampling for the reconstruction loss.

The reference calls `torch_geometric.utils.negative_sampling(pos ∪ self-loops, N)`
(dg_ae_model_aig.py:115-119): as many (src, dst) pairs as that edge set has, drawn uniformly from the
pairs that are neither an existing edge nor a self loop (pairs may cross graphs of a batch).  Same
distribution here, from rejection sampling on the device with torch index ops (the pairs feed the
HIP recon-loss kernel; SURVEY.md §8f row 3 lists a fused sampler as follow-up work)."""
import torch


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
