"""ORACLE — CPU restatement of the reference's DG_AE hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file;
the product (`multi-gate-vae_amd/`) never does and fails loudly when its HIP library is missing.

What it is: plain PyTorch-CPU fp32 arithmetic, written from the reference's source as a
*functional* restatement over a `{state_dict key: tensor}` dict (same key names and shapes as the
reference's `Model.state_dict()`), with the reference's per-node Python edge scans replaced by a
dst-sorted edge list.  Gradients come from torch autograd, the optimiser is `torch.optim.Adam`
(the reference's, `trainer.py:73`).

Parity pin: `tests/test_oracle_golden.py` checks every function here against the golden vectors in
`tests/golden/*.npz`, which `tests/golden/make_golden.py` produced by running the reference's own
`Model` / `Trainer.run_batch` / `DirectedGVAE.sample` code in the build container.

Reference files followed (all under /root/reference/DG_VAE/deepgate/):
  dg_ae_model_{aig,mig,xag,xmg}.py   Model.forward / pred_prob / recon_loss
  digae_layer.py:26-33,232-297        decoder, MultiGCNEncoder, DirectMultiGCNEncoder
  arch/gcn_conv.py:15-45              AggConv
  arch/tfmlp.py:11-52                 TFMlpAggr
  arch/mlp.py:14-56                   readout MLP
  trainer.py:131-174,229-234          run_batch, loss weighting
  utils/utils.py:32-36                zero_normalization
  digvae_model.py:134-142             reparameterisation sampler;  trainer.py:145-148  KL
"""
import torch
import torch.nn.functional as F

EPS = 1e-15  # dg_ae_model_aig.py:21

# gate id -> aggregator name, in the order each reference Model visits them inside a level
# (aig :73-95, mig :86-125, xag :92-120, xmg :97-145); the order is immaterial (disjoint node sets
# reading the pre-level state) but kept for readability.
GATES = {
    'aig': [(1, 'and'), (2, 'not')],
    'mig': [(2, 'not'), (3, 'and'), (4, 'or'), (1, 'maj')],
    'xag': [(3, 'and'), (2, 'not'), (5, 'xor')],
    'xmg': [(3, 'and'), (2, 'not'), (5, 'xor'), (1, 'maj'), (4, 'or')],
}
ENC_PREFIX = {'aig': 'struct_encoder', 'mig': 'mig_struct_encoder', 'xag': 'xag_struct_encoder',
              'xmg': 'xmg_struct_encoder'}


def linear(p, name, x):
    return F.linear(x, p[name + '.weight'], p[name + '.bias'])


def gru_cell(p, name, u, h):
    """torch.nn.GRU, one layer, seq_len 1 (gate order r,z,n)."""
    gi = F.linear(u, p[name + '.weight_ih_l0'], p[name + '.bias_ih_l0'])
    gh = F.linear(h, p[name + '.weight_hh_l0'], p[name + '.bias_hh_l0'])
    H = h.shape[-1]
    r = torch.sigmoid(gi[..., :H] + gh[..., :H])
    z = torch.sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    n = torch.tanh(gi[..., 2 * H:] + r * gh[..., 2 * H:])
    return (1 - z) * n + z * h


def agg_conv(p, name, h, src, dst):
    """AggConv (gcn_conv.py:30-45): out[i] = sum_{j->i} (W h_j + b); Linear applied per EDGE."""
    m = linear(p, name + '.msg', h.index_select(0, src))
    return torch.zeros_like(h).index_add_(0, dst, m)


def multi_gcn_encoder(p, name, x, edge_index, rounds, layernorm=True):
    """MultiGCNEncoder.forward (digae_layer.py:257-277): h0 = 1; per round a forward-edge half
    (aggr/update) and a reversed-edge half (aggr_r/update_r), the SAME LayerNorm after each."""
    N = x.shape[0]
    H = p[name + '.aggr.msg.weight'].shape[0]
    dt = p[name + '.aggr.msg.weight'].dtype          # fp32 like the reference; float64 parameters give a float64 restatement
    h = torch.ones(N, H, dtype=dt)
    xf = x.to(dt)
    src, dst = edge_index[0], edge_index[1]

    def ln(v):
        return F.layer_norm(v, (H,), p[name + '.ln.weight'], p[name + '.ln.bias']) if layernorm else v

    for _ in range(rounds):
        m = agg_conv(p, name + '.aggr', h, src, dst)
        h = ln(gru_cell(p, name + '.update', torch.cat([m, xf], dim=-1), h))
        m = agg_conv(p, name + '.aggr_r', h, dst, src)
        h = ln(gru_cell(p, name + '.update_r', torch.cat([m, xf], dim=-1), h))
    return h


def struct_encoder(p, prefix, x, edge_index, s_rounds, t_rounds, layernorm=True):
    """DirectMultiGCNEncoder.forward (digae_layer.py:294-297): two independent encoders, same edges."""
    s = multi_gcn_encoder(p, prefix + '.source_conv', x, edge_index, s_rounds, layernorm)
    t = multi_gcn_encoder(p, prefix + '.target_conv', x, edge_index, t_rounds, layernorm)
    return s, t


def segment_softmax(a, index, n):
    """torch_geometric.utils.softmax: exp(a - segmax) / (segsum + 1e-16)."""
    idx = index.view(-1, 1).expand_as(a)
    smax = a.new_full((n, a.shape[1]), float('-inf')).scatter_reduce(0, idx, a.detach(), reduce='amax')
    e = (a - smax.index_select(0, index)).exp()
    ssum = a.new_zeros((n, a.shape[1])).index_add_(0, index, e)
    return e / (ssum.index_select(0, index) + 1e-16)


def tf_mlp_aggr(p, name, x_src, x_dst, seg, n_seg):
    """TFMlpAggr.message + sum aggregation (tfmlp.py:38-46) for edges given as (x_j rows, x_i rows,
    destination segment id): q from x_i, k and v from x_j, attention over each destination's in-edges."""
    q = linear(p, name + '.msg_q', x_dst)
    k = linear(p, name + '.msg_k', x_src)
    a = linear(p, name + '.attn_lin', torch.cat([q, k], dim=-1))
    a = segment_softmax(a, seg, n_seg)
    v = linear(p, name + '.msg_v', x_src) * a
    return v.new_zeros((n_seg, v.shape[1])).index_add_(0, seg, v)


class LevelPlan:
    """dst-sorted edges + per-(level, gate) node lists: the vectorised stand-in for the reference's
    boolean masks (dg_ae_model_aig.py:72-75) and per-node `subgraph` scans (dag_utils.py:91-105)."""

    def __init__(self, ctype, edge_index, gate, forward_level):
        N = gate.shape[0]
        src, dst = edge_index[0], edge_index[1]
        order = torch.sort(dst, stable=True).indices
        self.src = src[order]
        self.dst = dst[order]
        deg = torch.bincount(dst, minlength=N)
        self.ptr = torch.zeros(N + 1, dtype=torch.long)
        self.ptr[1:] = torch.cumsum(deg, 0)
        g = gate.reshape(-1).to(torch.long)
        lv = forward_level.to(torch.long)
        self.num_levels = int(lv.max().item()) + 1 if N else 0
        self.groups = []   # (level, aggregator name, node ids, edge src ids, edge local segment)
        for level in range(1, self.num_levels):
            for gid, gname in GATES[ctype]:
                nodes = torch.nonzero((lv == level) & (g == gid)).reshape(-1)
                if nodes.numel() == 0:
                    continue
                cnt = deg[nodes]
                seg = torch.repeat_interleave(torch.arange(nodes.numel()), cnt)
                start = self.ptr[nodes]
                within = torch.arange(int(cnt.sum())) - torch.repeat_interleave(
                    torch.cumsum(cnt, 0) - cnt, cnt)
                eidx = torch.repeat_interleave(start, cnt) + within
                self.groups.append((level, gname, nodes, self.src[eidx], seg))


def model_forward(p, ctype, batch, s_rounds=4, t_rounds=4, layernorm=True, num_rounds=1, plan=None, fast=False):
    """Model.forward (dg_ae_model_aig.py:52-100 and siblings) -> (hs, hf, s, t)."""
    x = batch['x']
    N = x.shape[0]
    one_hot = F.one_hot(x[:, 1].to(torch.long), num_classes=6)      # the :59 quirk, kept
    s, t = struct_encoder(p, ENC_PREFIX[ctype], one_hot, batch['edge_index'], s_rounds, t_rounds, layernorm)
    hs = linear(p, 'hs_linear', torch.cat([s, t], dim=-1))
    H = hs.shape[1]
    hf = torch.zeros(N, H, dtype=hs.dtype)
    if plan is None:
        plan = LevelPlan(ctype, batch['edge_index'], batch['gate'], batch['forward_level'])
    if fast:
        assert num_rounds == 1
        return hs, sweep_fast(p, ctype, hs, plan), s, t
    for _ in range(num_rounds):
        level_writes = []
        cur = None
        for level, gname, nodes, esrc, seg in plan.groups:
            if cur is not None and level != cur:
                # node_state is refreshed once per level (:97): apply the level's writes now
                for nd, val in level_writes:
                    hf = hf.index_put((nd,), val)
                level_writes = []
            cur = level
            x_src = torch.cat([hs.index_select(0, esrc), hf.index_select(0, esrc)], dim=-1)
            dst_nodes = nodes.index_select(0, seg)
            x_dst = torch.cat([hs.index_select(0, dst_nodes), hf.index_select(0, dst_nodes)], dim=-1)
            msg = tf_mlp_aggr(p, 'aggr_%s_func' % gname, x_src, x_dst, seg, nodes.numel())
            hnew = gru_cell(p, 'update_%s_func' % gname, msg, hf.index_select(0, nodes))
            level_writes.append((nodes, hnew))
        for nd, val in level_writes:
            hf = hf.index_put((nd,), val)
    return hs, hf, s, t


def _group_update(p, gname, xs_s, xs_f, xd_s, xd_f, seg, n_seg):
    """TFMlpAggr + GRU of one (level, gate) group from a zero state (num_rounds = 1)."""
    msg = tf_mlp_aggr(p, 'aggr_%s_func' % gname, torch.cat([xs_s, xs_f], dim=-1), torch.cat([xd_s, xd_f], dim=-1), seg, n_seg)
    return gru_cell(p, 'update_%s_func' % gname, msg, msg.new_zeros((n_seg, msg.shape[1])))


class _SweepFast(torch.autograd.Function):
    """The level loop of `model_forward` with O(edges) cost.  The plain autograd version pays O(N) per
    (level, gate) group — `index_put` copies and dense index_select gradients, the same O(L*N) the
    reference pays with its per-level `torch.cat([hs, hf])` — which makes whole-workload CPU timing
    meaningless.  Here every group is still differentiated by torch autograd (locally, on the rows it
    touches); only the order of the groups and the accumulation into dhs/dhf are written out.
    `tests/test_oracle_golden.py` checks it against the plain version.  num_rounds = 1 only."""

    @staticmethod
    def forward(ctx, hs, plan, names, *params):
        p = dict(zip(names, params))
        hf = torch.zeros_like(hs)
        with torch.no_grad():
            for level, gname, nodes, esrc, seg in plan.groups:
                dn = nodes.index_select(0, seg)
                hf[nodes] = _group_update(p, gname, hs[esrc], hf[esrc], hs[dn], hf[dn], seg, nodes.numel())
        ctx.plan, ctx.names = plan, names
        ctx.save_for_backward(hs, hf, *params)
        return hf

    @staticmethod
    def backward(ctx, ghf):
        hs, hf, *params = ctx.saved_tensors
        names = ctx.names
        ghf = ghf.clone()
        ghs = torch.zeros_like(hs)
        gpar = [torch.zeros_like(t) for t in params]
        for level, gname, nodes, esrc, seg in reversed(ctx.plan.groups):
            dn = nodes.index_select(0, seg)
            leaves = [hs[esrc], hf[esrc], hs[dn], hf[dn]]
            used = [i for i, n in enumerate(names) if ('_%s_func' % gname) in n]
            local = [t.detach().requires_grad_(True) for t in leaves] + [params[i].detach().requires_grad_(True) for i in used]
            with torch.enable_grad():
                pl = {names[i]: local[4 + k] for k, i in enumerate(used)}
                out = _group_update(pl, gname, local[0], local[1], local[2], local[3], seg, nodes.numel())
            gs = torch.autograd.grad(out, local, grad_outputs=ghf[nodes], allow_unused=True)
            ghs.index_add_(0, esrc, gs[0]); ghf.index_add_(0, esrc, gs[1])
            if gs[2] is not None:
                ghs.index_add_(0, dn, gs[2])
            if gs[3] is not None:
                ghf.index_add_(0, dn, gs[3])
            for k, i in enumerate(used):
                if gs[4 + k] is not None:
                    gpar[i] += gs[4 + k]
        return (ghs, None, None, *gpar)


def sweep_fast(p, ctype, hs, plan):
    names = [k for k in p if k.startswith('aggr_') or k.startswith('update_')]
    return _SweepFast.apply(hs, plan, names, *[p[k] for k in names])


def readout_prob(p, hf, training, bn_state=None, p_drop=0.0, momentum=0.1, decisions=None):
    """pred_prob (dg_ae_model_aig.py:102-106) over MLP 64-32-32-1 with BatchNorm1d/ReLU/Dropout
    (mlp.py:27-47).  `bn_state` holds the running statistics (updated in place when training).
    `decisions` (checker aid, not in the reference): {'relu': [mask1, mask2], 'inside': mask} — the piecewise-linear branches another
    run took (which ReLUs passed, which outputs the clamp left alone), imposed here instead of decided here, so that two runs whose
    pre-activations differ by rounding are compared on the SAME linear piece."""
    name = 'readout_prob.fc'
    y = hf
    for blk, (lin, bn) in enumerate(((0, 1), (4, 5))):
        y = linear(p, '%s.%d' % (name, lin), y)
        rm = p['%s.%d.running_mean' % (name, bn)] if bn_state is None else bn_state['%s.%d.running_mean' % (name, bn)]
        rv = p['%s.%d.running_var' % (name, bn)] if bn_state is None else bn_state['%s.%d.running_var' % (name, bn)]
        y = F.batch_norm(y, rm, rv, p['%s.%d.weight' % (name, bn)], p['%s.%d.bias' % (name, bn)],
                         training=training, momentum=momentum, eps=1e-5)
        y = F.relu(y) if decisions is None else y * decisions['relu'][blk].to(y.dtype)
        y = F.dropout(y, p_drop, training=training)
    y = linear(p, name + '.8', y)
    if decisions is not None:
        return torch.where(decisions['inside'], y, torch.clamp(y, min=0.0, max=1.0).detach())
    return torch.clamp(y, min=0.0, max=1.0)


def decoder(s, t, edge_index, sigmoid=True):
    """DirectedInnerProductDecoder.forward (digae_layer.py:27-29)."""
    v = (s.index_select(0, edge_index[0]) * t.index_select(0, edge_index[1])).sum(dim=1)
    return torch.sigmoid(v) if sigmoid else v


def recon_loss(p, hs, pos_edge_index, neg_edge_index):
    """Model.recon_loss (dg_ae_model_aig.py:108-130) with explicit negatives."""
    st = linear(p, 'hs_decompose', hs)
    H = hs.shape[1]
    s, t = st[:, :H], st[:, H:]
    pos = decoder(s, t, pos_edge_index)
    neg = decoder(s, t, neg_edge_index)
    loss = -torch.log(pos + EPS).mean() - torch.log(1 - neg + EPS).mean()
    pred_bin = torch.cat([(pos > 0.5), (neg > 0.5)]).to(torch.int32)
    gt_bin = torch.cat([torch.ones_like(pos), torch.zeros_like(neg)]).to(torch.int32)
    return loss, pred_bin, gt_bin


def zero_normalization(x):
    """utils/utils.py:32-36 (unbiased std)."""
    return (x - x.mean()) / x.std()


def func_loss(hf, tt_pair_index, tt_sim):
    """trainer.py:158-163."""
    a = hf.index_select(0, tt_pair_index[0])
    b = hf.index_select(0, tt_pair_index[1])
    dis = 1 - F.cosine_similarity(a, b, eps=1e-8)
    return F.l1_loss(zero_normalization(dis), zero_normalization(tt_sim)), dis


def run_batch(p, ctype, batch, training=True, bn_state=None, p_drop=0.0, s_rounds=4, t_rounds=4,
              layernorm=True, num_rounds=1, plan=None, fast=False, decisions=None):
    """Trainer.run_batch (trainer.py:131-174).  The edge split keeps only its live effect — a
    permutation of the edges, to which the mean over edges is invariant — and never builds the dead
    N x N mask (preprocessing.py:56-69)."""
    hs, hf, s, t = model_forward(p, ctype, batch, s_rounds, t_rounds, layernorm, num_rounds, plan, fast)
    rl, pred_bin, gt_bin = recon_loss(p, hs, batch['edge_index'], batch['neg_edge_index'])
    prob = readout_prob(p, hf, training, bn_state, p_drop, decisions=decisions)
    # (`decisions`: see readout_prob; 'sign' = the L1 loss's branch per node, imposed)
    pl = F.l1_loss(prob, batch['prob']) if decisions is None else (decisions['sign'].to(prob.dtype) * (prob - batch['prob'].to(prob.dtype))).mean()
    fl, _ = func_loss(hf, batch['tt_pair_index'], batch['tt_sim'])
    return {'recon_loss': rl, 'pred_bin': pred_bin, 'gt_bin': gt_bin, 'prob_loss': pl, 'func_loss': fl,
            'hs': hs, 'hf': hf, 's': s, 't': t, 'prob': prob}


def weighted_loss(ls, w):
    """trainer.py:229-231."""
    return w[0] * ls['recon_loss'] + w[1] * ls['prob_loss'] + w[2] * ls['func_loss']


def confusion(pred_bin, gt_bin):
    """trainer.py:240-244: acc, TP, FP, TN, FN as fractions of len(pred_bin)."""
    n = float(pred_bin.numel())
    pb, gb = pred_bin.to(torch.long), gt_bin.to(torch.long)
    return {'acc': float((pb == gb).sum()) / n, 'TP': float(((pb == 1) & (gb == 1)).sum()) / n,
            'FP': float(((pb == 1) & (gb == 0)).sum()) / n, 'TN': float(((pb == 0) & (gb == 0)).sum()) / n,
            'FN': float(((pb == 0) & (gb == 1)).sum()) / n}


def vae_sample(p, s, t, eps_s, eps_t):
    """DirectedGVAE.sample (digvae_model.py:134-142) with the two randn_like draws injected."""
    s_mu, s_ls = linear(p, 'fc_s_mu', s), linear(p, 'fc_s_logstd', s)
    t_mu, t_ls = linear(p, 'fc_t_mu', t), linear(p, 'fc_t_logstd', t)
    return s_mu + torch.exp(s_ls) * eps_s, t_mu + torch.exp(t_ls) * eps_t, (s_mu, s_ls, t_mu, t_ls)


def kl_term(mu, logstd):
    """trainer.py:146-147 exactly as written (note the double 1/N)."""
    n = mu.shape[0]
    return -0.5 / n * (1 + 2 * logstd - mu ** 2 - torch.exp(logstd) ** 2).sum(1).mean()


# ---------------------------------------------------------------------------- helpers for callers
def params_from_npz(z, prefix='param_', requires_grad=True):
    """{key: tensor} from a golden .npz; float tensors become autograd leaves."""
    out = {}
    for k in z.files:
        if k.startswith(prefix):
            v = torch.from_numpy(z[k].copy())
            if v.is_floating_point() and requires_grad and 'running_' not in k:
                v.requires_grad_(True)
            out[k[len(prefix):]] = v
    return out


def batch_from_arrays(get):
    """Batch dict of tensors from a mapping/function name -> numpy array (golden 'in_' arrays or a
    synthetic batch)."""
    b = {}
    for k in ('x', 'edge_index', 'gate', 'forward_level', 'forward_index', 'prob', 'tt_pair_index',
              'tt_sim', 'neg_edge_index'):
        b[k] = torch.from_numpy(get(k).copy())
    return b


def trainable(p):
    return [v for k, v in p.items() if v.requires_grad]
