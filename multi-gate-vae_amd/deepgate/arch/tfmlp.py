"""TFMlpAggr (reference: DG_VAE/deepgate/arch/tfmlp.py:11-52): per-destination attention over the
in-edges, q from the destination, k/v from the sources.

Inside a Model the module is a parameter container: `Model.forward` composes `attn_u`, `Wvc`, `bvc`
from its weights and runs the levelised sweep kernel.  `composed()` is that composition; `forward()` (an edge-list
call on its own) runs the attention-pooling kernels."""
import torch
import torch.nn as nn


class TFMlpAggr(nn.Module):
    def __init__(self, in_channels, ouput_channels=64, reverse=False, mlp_post=None):
        super().__init__()
        if ouput_channels is None:
            ouput_channels = in_channels
        assert (in_channels > 0) and (ouput_channels > 0), 'The dimension for the DeepSetConv should be larger than 0.'
        if mlp_post is not None or reverse:
            raise NotImplementedError('mlp_post / reverse are not used by the DG_AE path')
        self.msg_post = None
        self.attn_lin = nn.Linear(ouput_channels + ouput_channels, 1)
        self.msg_q = nn.Linear(in_channels, ouput_channels)
        self.msg_k = nn.Linear(in_channels, ouput_channels)
        self.msg_v = nn.Linear(in_channels, ouput_channels)

    def composed(self, gru):
        """(attn_u [in], Wvc [3H,in], bvc [3H], b_ih, b_hh) for the sweep kernel.

        attention score of edge j->i = w_q.(Wq x_i + bq) + w_k.(Wk x_j + bk) + b: everything but
        (Wk^T w_k).x_j is constant over i's softmax segment and cancels, so msg_q, msg_k.bias,
        attn_lin.bias and the q half of attn_lin.weight never influence the output (their reference
        gradients are rounding noise, ~1e-9); the value Linear is folded into the GRU input side."""
        out = self.msg_k.weight.shape[0]
        w_k = self.attn_lin.weight[0, out:]
        attn_u = w_k @ self.msg_k.weight
        w_ih = gru.weight_ih_l0
        return attn_u, w_ih @ self.msg_v.weight, w_ih @ self.msg_v.bias, gru.bias_ih_l0, gru.bias_hh_l0

    def forward(self, x, edge_index, edge_attr=None, plan=None, **kwargs):
        """Stand-alone edge-list call (tfmlp.py:31-35): [N, out] messages, zero rows for nodes without in-edges.
        Attention pooling of the source rows (csrc/attn_pool.hip: the q term is constant over a destination's softmax segment and
        cancels, so the score is (Wk^T w_k).x_j, tfmlp.py:38-46) followed by the value Linear,
        W_v (sum_j alpha_j x_j) + b_v [deg > 0] = sum_j alpha_j (W_v x_j + b_v).  `plan` = a GraphPlan of edge_index saves building
        the CSR.  Device tensors only: there is no CPU implementation."""
        from .. import ops
        from ..graph_plan import GraphPlan
        if not x.is_cuda:
            raise ops._hip.HipLibraryError('TFMlpAggr.forward needs device tensors (got %s); there is no CPU implementation' % x.device)
        if plan is None:
            plan = GraphPlan(edge_index, x.shape[0])
        out = self.msg_k.weight.shape[0]
        u = self.attn_lin.weight[0, out:] @ self.msg_k.weight
        zbar = ops.AttnPoolFn.apply(x, u, plan)
        has_in = (plan.in_ptr[1:] > plan.in_ptr[:-1]).to(x.dtype).unsqueeze(1)
        return ops.linear(zbar, self.msg_v.weight) + has_in * self.msg_v.bias
