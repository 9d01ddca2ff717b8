"""TFMlpAggr (reference: DG_VAE/deepgate/arch/tfmlp.py:11-52): per-destination attention over the
in-edges, q from the destination, k/v from the sources.

Inside a Model the module is a parameter container: `Model.forward` composes `attn_u`, `Wvc`, `bvc`
from its weights and runs the levelised sweep kernel.  `composed()` is that composition."""
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

    def forward(self, x, edge_index, edge_attr=None, **kwargs):
        raise NotImplementedError('TFMlpAggr runs as part of Model.forward (levelised sweep kernel); '
                                  'a stand-alone edge-list call is not on the DG_AE path')
