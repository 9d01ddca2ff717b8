import torch


def softmax(src, index, ptr=None, num_nodes=None, dim=0):
    """PyG segment softmax: exp(src - segmax) / (segsum + 1e-16)."""
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    shape = [n] + list(src.shape[1:])
    idx = index.view([-1] + [1] * (src.dim() - 1)).expand_as(src)
    smax = src.new_full(shape, float('-inf')).scatter_reduce(0, idx, src, reduce='amax', include_self=True)
    out = (src - smax.index_select(0, index)).exp()
    ssum = src.new_zeros(shape).index_add_(0, index, out)
    return out / (ssum.index_select(0, index) + 1e-16)


def degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros(n, dtype=dtype or torch.float)
    return out.index_add_(0, index, torch.ones_like(index, dtype=out.dtype))


def remove_self_loops(edge_index, edge_attr=None):
    keep = edge_index[0] != edge_index[1]
    return edge_index[:, keep], None


def add_self_loops(edge_index, edge_attr=None, fill_value=None, num_nodes=None):
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    loop = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index, torch.stack([loop, loop])], dim=1), None


def negative_sampling(edge_index, num_nodes=None, num_neg_samples=None, **k):
    raise RuntimeError('negative_sampling stand-in: golden generation always passes neg_edge_index')


def to_undirected(edge_index, *a, **k):
    raise NotImplementedError
