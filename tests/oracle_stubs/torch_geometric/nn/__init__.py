import inspect
import torch
from . import glob  # noqa: F401


class MessagePassing(torch.nn.Module):
    """propagate(): gather per-edge rows, call message(), sum into destinations, call update()."""

    def __init__(self, aggr='add', flow='source_to_target', node_dim=-2):
        super().__init__()
        assert aggr == 'add'
        self.flow = flow
        self.node_dim = node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        j, i = (0, 1) if self.flow == 'source_to_target' else (1, 0)
        names = list(inspect.signature(self.message).parameters)
        n_nodes = None
        for v in kwargs.values():
            if torch.is_tensor(v):
                n_nodes = v.size(self.node_dim)
                break
        feed = {}
        for name in names:
            if name.endswith('_j') and name[:-2] in kwargs:
                feed[name] = kwargs[name[:-2]].index_select(self.node_dim, edge_index[j])
            elif name.endswith('_i') and name[:-2] in kwargs and name != 'size_i':
                feed[name] = kwargs[name[:-2]].index_select(self.node_dim, edge_index[i])
            elif name == 'index':
                feed[name] = edge_index[i]
            elif name == 'ptr':
                feed[name] = None
            elif name == 'size_i':
                feed[name] = n_nodes
            else:
                feed[name] = kwargs.get(name)
        msg = self.message(**feed)
        shape = list(msg.shape)
        shape[self.node_dim] = n_nodes
        out = msg.new_zeros(shape)
        out.index_add_(self.node_dim if self.node_dim >= 0 else msg.dim() + self.node_dim, edge_index[i], msg)
        return self.update(out)

    def message(self, x_j):
        return x_j

    def update(self, aggr_out):
        return aggr_out
