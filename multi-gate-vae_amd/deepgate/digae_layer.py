"""Structural encoder / decoder surface of the reference (`DG_VAE/deepgate/digae_layer.py:26-33,
232-297`) on top of the HIP kernels.  Module and parameter names, shapes, construction order
(hence seeded initialisation) and call signatures follow the reference, so its checkpoints load and
`train.py` builds the encoder the same way; the arithmetic is `ops.StructEncoderFn`.
"""
import torch
import torch.nn as nn

from . import ops
from .arch.gcn_conv import AggConv
from .graph_plan import GraphPlan

MAX_FEATURE_CLASSES = 8


def feature_classes(x):
    """Distinct rows of the node-feature matrix and each node's row id.  The kernels add the GRU's
    feature term as a per-class table (`W_ih[:, H:] x_c + b_ih`), exact for any x with few distinct
    rows — the reference always feeds one-hot rows (dg_ae_model_aig.py:59)."""
    rows, inv = torch.unique(x, dim=0, return_inverse=True)
    if rows.shape[0] > MAX_FEATURE_CLASSES:
        return None                          # general features: MultiGCNEncoder.forward forms the term per node (_forward_rows)
    return rows.to(torch.float32), inv.to(torch.uint8).contiguous()


class DirectedInnerProductDecoder(nn.Module):
    """sigma(<s[src], t[dst]>) per edge (digae_layer.py:26-33)."""

    def forward(self, s, t, edge_index, sigmoid=True):
        return ops.edge_dot(s, t, edge_index, sigmoid)

    def forward_all(self, s, t, sigmoid=True):
        # dense N x N scores: only sensible for tiny graphs; not on the training path
        adj = ops.dense_scores(s, t)
        return torch.sigmoid(adj) if sigmoid else adj


class MultiGCNEncoder(nn.Module):
    def __init__(self, num_rounds, dim_hidden, dim_feature, enable_reverse, layernorm):
        super().__init__()
        self.num_rounds = num_rounds
        self.enable_reverse = True          # the reference forces this on (digae_layer.py:238)
        self.layernorm = layernorm
        self.dim_feature = dim_feature
        self.dim_hidden = dim_hidden
        self.aggr = AggConv(dim_hidden, dim_hidden)
        self.update = nn.GRU(dim_hidden + dim_feature, dim_hidden)
        self.aggr_r = AggConv(dim_hidden, dim_hidden)
        self.update_r = nn.GRU(dim_hidden + dim_feature, dim_hidden)
        if self.layernorm:
            self.ln = nn.LayerNorm(dim_hidden)

    def _composed(self, aggr, gru, feat_rows):
        """Fold the per-edge message Linear into the GRU input projection (tiny weight-space
        products; autograd differentiates them)."""
        H = self.dim_hidden
        w_ih = gru.weight_ih_l0
        w_m = w_ih[:, :H]
        Wc = w_m @ aggr.msg.weight
        bc = w_m @ aggr.msg.bias
        xtab = feat_rows @ w_ih[:, H:].t() + gru.bias_ih_l0
        return xtab, Wc, bc, gru.weight_hh_l0, gru.bias_hh_l0

    def _forward_rows(self, x, edge_index, plan):
        """General node features (more distinct rows than the class table holds; digae_layer.py:257-277 takes any x [N, F]): the GRU's
        feature term W_ih[:, H:] x_i + b_ih per node through the linear kernels (x zero-padded to their 16-column granule), every
        half round per node on the exact-fp32 stage kernels (ops.StructEncoderRowsFn)."""
        H, F_ = self.dim_hidden, self.dim_feature
        if plan is None:
            plan = GraphPlan(edge_index, x.shape[0])
        pad = (-F_) % 16
        xp = torch.nn.functional.pad(x.to(torch.float32), (0, pad)).contiguous()
        args = []
        for aggr, gru in ((self.aggr, self.update), (self.aggr_r, self.update_r)):
            w_ih = gru.weight_ih_l0
            w_m = w_ih[:, :H]
            w_x = torch.nn.functional.pad(w_ih[:, H:], (0, pad))
            # one Linear per gate block (the linear kernels serve M = H outputs, not 3H), side by side
            xrow = torch.cat([ops.linear(xp, w_x[g * H:(g + 1) * H], gru.bias_ih_l0[g * H:(g + 1) * H]) for g in range(3)], dim=1)
            args += [xrow, w_m @ aggr.msg.weight, w_m @ aggr.msg.bias, gru.weight_hh_l0, gru.bias_hh_l0]
        ln_w = self.ln.weight if self.layernorm else None
        ln_b = self.ln.bias if self.layernorm else None
        return ops.StructEncoderRowsFn.apply(plan, self.num_rounds, *args, ln_w, ln_b)

    def forward(self, x, edge_index, plan=None, classes=None):
        """`classes` = (distinct feature rows [C,F], row id per node uint8 [N]) may stand in for x."""
        if classes is None:
            if x.shape[1] != self.dim_feature:
                raise ValueError('expected %d node features, got %d' % (self.dim_feature, x.shape[1]))
            classes = feature_classes(x)
        if classes is None:
            return self._forward_rows(x, edge_index, plan)
        rows, xcls = classes
        if rows.shape[1] != self.dim_feature:
            raise ValueError('expected %d node features, got %d' % (self.dim_feature, rows.shape[1]))
        if plan is None:
            plan = GraphPlan(edge_index, xcls.shape[0])
        rows = rows.to(self.update.weight_ih_l0.device)
        f = self._composed(self.aggr, self.update, rows)
        r = self._composed(self.aggr_r, self.update_r, rows)
        ln_w = self.ln.weight if self.layernorm else None
        ln_b = self.ln.bias if self.layernorm else None
        return ops.StructEncoderFn.apply(plan, xcls, self.num_rounds, *f, *r, ln_w, ln_b)


class DirectMultiGCNEncoder(nn.Module):
    def __init__(self, dim_feature=3, dim_hidden=128, s_rounds=1, t_rounds=1, enable_reverse=True, layernorm=False):
        super().__init__()
        self.source_conv = MultiGCNEncoder(s_rounds, dim_hidden, dim_feature, enable_reverse, layernorm)
        self.target_conv = MultiGCNEncoder(t_rounds, dim_hidden, dim_feature, enable_reverse, layernorm)

    def forward(self, s, t, edge_index, plan=None, classes=None):
        if plan is None:
            plan = GraphPlan(edge_index, s.shape[0])
        cs = classes if classes is not None else feature_classes(s)
        ct = cs if (classes is not None or t is s) else feature_classes(t)
        # (None: more distinct feature rows than the class table holds -> the per-node feature path of MultiGCNEncoder)
        return self.source_conv(s, edge_index, plan, cs), self.target_conv(t, edge_index, plan, ct)
