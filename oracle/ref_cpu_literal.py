"""ORACLE, literal variant — CPU restatement of the reference's DG_AE train step INCLUDING the parts that make the
reference itself slow.  TEST INFRASTRUCTURE, NOT PRODUCT (same rules as oracle/ref_cpu.py: only tests/, smoke() and bench.py's
cpu_baseline leg may import it).

`oracle/ref_cpu.py` replaces three O(N*E) / O(N^2) constructions of the reference by O(E) ones with identical results.  This file
keeps them as the reference has them, so that timing it at BASELINE config 1 ties back to the survey's measurement of the real
reference (1.5-1.8 s/step on 8 cores) on whatever host the benchmark runs on:

  * `subgraph` — one boolean scan of ALL edges per target node (utils/dag_utils.py:91-105), called once per (level, gate type)
    group (dg_ae_model_aig.py:76,87);
  * the level loop — boolean level/gate masks over all nodes, TFMlpAggr scattering into a full [N, H] buffer, `index_select`
    of the level's rows, in-place `hf[nodes] = ...`, and `torch.cat([hs, hf])` of the whole state after every level
    (dg_ae_model_aig.py:70-97);
  * `general_train_test_split_edges` — the edge permutation plus the dead dense N x N uint8/bool mask, its `nonzero` and the
    `randperm` over the non-edges (preprocessing.py:41-69).

Everything else (encoder, GRU, attention, readout, losses) is oracle/ref_cpu.py's code.  Parity pin:
tests/test_oracle_golden.py::test_literal_oracle_* (against ref_cpu and against the g1 fixtures of the reference).
Feasible only at small N: the mask alone is N^2 bytes (17.6 TB at config 2).
"""
import torch
import torch.nn.functional as F

from . import ref_cpu as R


def subgraph(target_idx, edge_index, dim=1):
    """utils/dag_utils.py:91-105: edges whose endpoint `dim` is in target_idx, concatenated in target order."""
    le_idx = []
    for n in target_idx:
        ne_idx = edge_index[dim] == n
        le_idx += [ne_idx.nonzero().squeeze(-1)]
    le_idx = torch.cat(le_idx, dim=-1)
    return edge_index[:, le_idx]


def general_train_test_split_edges(edge_index, num_nodes):
    """preprocessing.py:41-69 with val_ratio = test_ratio = 0, directed: returns train_pos_edge_index (a permutation of the
    edges); builds — and drops, as the reference does — the dense negative-adjacency mask."""
    row, col = edge_index
    perm = torch.randperm(row.size(0))
    row, col = row[perm], col[perm]
    train_pos_edge_index = torch.stack([row, col], dim=0)
    neg_adj_mask = torch.ones(num_nodes, num_nodes, dtype=torch.uint8)
    neg_adj_mask = neg_adj_mask.to(torch.bool)
    neg_adj_mask[row, col] = 0
    neg_row, neg_col = neg_adj_mask.nonzero(as_tuple=False).t()
    perm = torch.randperm(neg_row.size(0))[:0]
    neg_row, neg_col = neg_row[perm], neg_col[perm]
    neg_adj_mask[neg_row, neg_col] = 0
    return train_pos_edge_index


def _aggr_full(p, name, node_state, sub_ei):
    """TFMlpAggr.forward over the whole node set (tfmlp.py:31-46): messages of the sub-edges, summed into an [N, H] buffer."""
    n = node_state.shape[0]
    return R.tf_mlp_aggr(p, name, node_state.index_select(0, sub_ei[0]), node_state.index_select(0, sub_ei[1]), sub_ei[1], n)


def model_forward(p, ctype, batch, s_rounds=4, t_rounds=4, layernorm=True, num_rounds=1):
    """Model.forward as written in dg_ae_model_aig.py:52-100 (siblings: the same loop over their gate sets)."""
    x, edge_index = batch['x'], batch['edge_index']
    num_nodes = x.shape[0]
    forward_level = batch['forward_level']
    forward_index = batch['forward_index']
    num_layers_f = int(forward_level.max().item()) + 1
    one_hot = F.one_hot(x[:, 1].to(torch.long), num_classes=6)
    s, t = R.struct_encoder(p, R.ENC_PREFIX[ctype], one_hot, edge_index, s_rounds, t_rounds, layernorm)
    hs = R.linear(p, 'hs_linear', torch.cat([s, t], dim=-1))
    H = hs.shape[1]
    hf = torch.zeros(num_nodes, H)
    node_state = torch.cat([hs, hf], dim=-1)
    gate = batch['gate'].reshape(-1)
    masks = [(gname, gate == gid) for gid, gname in R.GATES[ctype]]
    for _ in range(num_rounds):
        for level in range(1, num_layers_f):
            layer_mask = forward_level == level
            for gname, gmask in masks:
                l_node = forward_index[layer_mask & gmask]
                if l_node.size(0) > 0:
                    sub_ei = subgraph(l_node, edge_index, dim=1)
                    msg = _aggr_full(p, 'aggr_%s_func' % gname, node_state, sub_ei)
                    g_msg = torch.index_select(msg, dim=0, index=l_node)
                    hf_g = torch.index_select(hf, dim=0, index=l_node)
                    hf_new = R.gru_cell(p, 'update_%s_func' % gname, g_msg, hf_g)
                    hf = hf.index_put((l_node,), hf_new)           # `hf[l_node, :] = ...` without breaking autograd's versioning
            node_state = torch.cat([hs, hf], dim=-1)
    return hs, hf, s, t


def run_batch(p, ctype, batch, training=True, bn_state=None, p_drop=0.0, s_rounds=4, t_rounds=4, layernorm=True, num_rounds=1):
    """Trainer.run_batch (trainer.py:131-174) with the edge split in front, as the reference calls it (:133)."""
    pos = general_train_test_split_edges(batch['edge_index'], batch['x'].shape[0])
    hs, hf, s, t = model_forward(p, ctype, batch, s_rounds, t_rounds, layernorm, num_rounds)
    rl, pred_bin, gt_bin = R.recon_loss(p, hs, pos, batch['neg_edge_index'])
    prob = R.readout_prob(p, hf, training, bn_state, p_drop)
    pl = F.l1_loss(prob, batch['prob'])
    fl, _ = R.func_loss(hf, batch['tt_pair_index'], batch['tt_sim'])
    return {'recon_loss': rl, 'pred_bin': pred_bin, 'gt_bin': gt_bin, 'prob_loss': pl, 'func_loss': fl,
            'hs': hs, 'hf': hf, 's': s, 't': t, 'prob': prob}
