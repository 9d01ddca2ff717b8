"""AggConv (reference: DG_VAE/deepgate/arch/gcn_conv.py:15-45): out[i] = sum_{j->i} msg(x[j]).

Inside the structural encoder the module is a parameter container — `MultiGCNEncoder` folds
`msg` into the GRU input projection and runs the fused half-round kernel.  Called on its own it runs
the same arithmetic through the gather-sum and Linear kernels."""
import torch.nn as nn

from .. import ops
from ..graph_plan import GraphPlan


class AggConv(nn.Module):
    def __init__(self, in_channels, ouput_channels=None, wea=False, mlp=None, reverse=False):
        super().__init__()
        if ouput_channels is None:
            ouput_channels = in_channels
        assert (in_channels > 0) and (ouput_channels > 0), 'The dimension for the AggConv should be larger than 0.'
        if wea or mlp is not None:
            raise NotImplementedError('edge attributes / custom message MLPs are not used by the DG_AE path')
        self.wea = wea
        self.reverse = reverse
        self.msg = nn.Linear(in_channels, ouput_channels)

    def forward(self, x, edge_index, edge_attr=None, plan=None, **kwargs):
        squeeze = x.dim() == 3          # the encoder passes [1, N, H] (node_dim = -2)
        h = x[0] if squeeze else x
        if plan is None:
            plan = GraphPlan(edge_index, h.shape[0])
        ptr, idx = plan.csr(self.reverse)
        agg, deg = ops.gather_sum(h, ptr, idx)
        out = ops.linear(agg, self.msg.weight, None) + deg.unsqueeze(1) * self.msg.bias
        return out.unsqueeze(0) if squeeze else out
