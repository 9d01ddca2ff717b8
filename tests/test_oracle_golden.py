"""Pin the oracle (oracle/ref_cpu.py) to the golden vectors the reference itself produced
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import ref_cpu as R

TYPES = ['aig', 'mig', 'xag', 'xmg']


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def _model_case(z):
    ctype = str(z['meta_type'])
    Rr = int(z['meta_R'])
    p = R.params_from_npz(z)
    batch = R.batch_from_arrays(lambda k: z['in_' + k])
    return ctype, Rr, p, batch


@pytest.mark.parametrize('name', ['g1_' + t for t in TYPES] + ['g2_aig', 'g2_xmg'])
def test_eval_forward_matches_reference(name):
    z = load(name)
    ctype, Rr, p, batch = _model_case(z)
    with torch.no_grad():
        hs, hf, s, t = R.model_forward(p, ctype, batch, Rr, Rr)
        prob = R.readout_prob(p, hf, training=False)
        rl, pred_bin, gt_bin = R.recon_loss(p, hs, batch['edge_index'], batch['neg_edge_index'])
    close(s, z["eval_s"], 5e-5, 2e-5)
    close(t, z["eval_t"], 5e-5, 2e-5)
    close(hs, z["eval_hs"], 5e-5, 2e-5)
    close(hf, z["eval_hf"], 5e-5, 2e-5)
    close(prob, z["eval_prob"], 5e-5, 2e-5)
    close(rl, z['eval_recon'], 1e-5)
    assert np.array_equal(pred_bin.numpy(), z['eval_pred_bin'])
    assert np.array_equal(gt_bin.numpy(), z['eval_gt_bin'])


@pytest.mark.parametrize('name', ['g1_' + t for t in TYPES] + ['g2_aig', 'g2_xmg'])
def test_train_losses_grads_and_adam_match_reference(name):
    z = load(name)
    ctype, Rr, p, batch = _model_case(z)
    bn = {k: v.clone() for k, v in p.items() if 'running_' in k}
    ls = R.run_batch(p, ctype, batch, training=True, bn_state=bn, p_drop=0.0, s_rounds=Rr, t_rounds=Rr)
    w = z['meta_weights']
    loss = R.weighted_loss(ls, w)
    close(ls['recon_loss'], z['train_recon'], 1e-5)
    close(ls['prob_loss'], z['train_prob_loss'], 1e-5)
    close(ls['func_loss'], z['train_func_loss'], 1e-5)
    close(loss, z['train_loss'], 1e-5)
    assert int(ls['pred_bin'].sum()) == int(z['train_pred_bin_sum'])
    assert int(ls['gt_bin'].sum()) == int(z['train_gt_bin_sum'])
    names = [k for k, v in p.items() if v.requires_grad]
    opt = torch.optim.Adam([p[k] for k in names], lr=float(z['meta_lr']))
    loss.backward()
    for k in names:
        g = p[k].grad if p[k].grad is not None else torch.zeros_like(p[k])
        ref = z['grad_' + k]
        # 1e-6 floor: parameters whose gradient is mathematically zero (Linear biases in front of a
        # BatchNorm, the q branch of TFMlpAggr) carry only rounding noise of that size in the reference
        scale = max(1e-6, float(np.abs(ref).max()))
        np.testing.assert_allclose(g.numpy(), ref, rtol=2e-4, atol=2e-5 * scale + 1e-6, err_msg=k)
    if 'after_hs_linear.weight' in z.files:
        opt.step()
        for k in names:
            # Adam's first step is lr*sign(g): elements whose reference gradient is rounding noise
            # move by +-lr in a noise-determined direction, so they are excluded
            live = np.abs(z['grad_' + k]) > 1e-5
            np.testing.assert_allclose(p[k].detach().numpy()[live], z['after_' + k][live], rtol=1e-5, atol=2e-6, err_msg=k)
        for k, v in bn.items():
            close(v, z['after_' + k], 1e-5, 1e-6)


def test_encoder_half_rounds():
    z = load('g3_ops')
    p = R.params_from_npz(z, 'enc_param_')
    p = {'e.' + k: v for k, v in p.items()}
    ei = torch.from_numpy(z['enc_edge_index'])
    x = torch.from_numpy(z['enc_x']).float()
    h0 = torch.from_numpy(z['enc_h0']).requires_grad_(True)
    H = h0.shape[1]
    m = R.agg_conv(p, 'e.aggr', h0, ei[0], ei[1])
    close(m, z['enc_msg'], 1e-5, 1e-5)
    h1p = R.gru_cell(p, 'e.update', torch.cat([m, x], -1), h0)
    close(h1p, z['enc_h1_pre_ln'], 1e-5, 1e-5)
    ln = lambda v: torch.nn.functional.layer_norm(v, (H,), p['e.ln.weight'], p['e.ln.bias'])
    h1 = ln(h1p)
    close(h1, z['enc_h1'], 1e-5, 1e-5)
    m2 = R.agg_conv(p, 'e.aggr_r', h1, ei[1], ei[0])
    h2 = ln(R.gru_cell(p, 'e.update_r', torch.cat([m2, x], -1), h1))
    close(h2, z['enc_h2'], 1e-5, 1e-5)
    (h2 * torch.from_numpy(z['enc_up'])).sum().backward()
    close(h0.grad, z['enc_grad_h0'], 1e-4, 1e-5)
    for k, v in p.items():
        close(v.grad, z['enc_grad_' + k[2:]], 1e-4, 1e-4)


def test_functional_level():
    z = load('g3_ops')
    p = {'a.' + k: v for k, v in R.params_from_npz(z, 'lvl_aggr_').items()}
    p.update({'g.' + k: v for k, v in R.params_from_npz(z, 'lvl_gru_').items()})
    ei = torch.from_numpy(z['lvl_edge_index'])
    nodes = torch.from_numpy(z['lvl_nodes'])
    ns = torch.from_numpy(z['lvl_node_state']).requires_grad_(True)
    hp = torch.from_numpy(z['lvl_hprev']).requires_grad_(True)
    keep = torch.isin(ei[1], nodes)
    src, dst = ei[0][keep], ei[1][keep]
    order = torch.sort(dst, stable=True).indices
    src, dst = src[order], dst[order]
    seg = torch.searchsorted(nodes, dst)
    msg = R.tf_mlp_aggr(p, 'a', ns[src], ns[dst], seg, nodes.numel())
    close(msg, z['lvl_msg'], 1e-5, 1e-5)
    hn = R.gru_cell(p, 'g', msg, hp[nodes])
    close(hn, z['lvl_hnew'], 1e-5, 1e-5)
    (hn * torch.from_numpy(z['lvl_up'])).sum().backward()
    close(ns.grad, z['lvl_grad_node_state'], 1e-4, 1e-5)
    close(hp.grad, z['lvl_grad_hprev'], 1e-4, 1e-5)
    for k, v in p.items():
        ref = z['lvl_grad_' + ('aggr_' if k.startswith('a.') else 'gru_') + k[2:]]
        close(v.grad, ref, 1e-4, 2e-5)


def test_readout_batchnorm_train_mode():
    z = load('g3_ops')
    p = {'readout_prob.' + k: v for k, v in R.params_from_npz(z, 'mlp_param_').items()}
    # the fixture stores the state_dict AFTER the training-mode forward: rebuild the "before" buffers
    bn = {}
    for b in (1, 5):
        bn['readout_prob.fc.%d.running_mean' % b] = torch.zeros(32)
        bn['readout_prob.fc.%d.running_var' % b] = torch.ones(32)
    x = torch.from_numpy(z['mlp_in']).requires_grad_(True)
    pr = R.readout_prob(p, x, training=True, bn_state=bn, p_drop=0.0)
    close(pr, z['mlp_prob'], 1e-5, 1e-6)
    l1 = torch.nn.functional.l1_loss(pr, torch.from_numpy(z['mlp_target']))
    close(l1, z['mlp_l1'], 1e-5)
    l1.backward()
    close(x.grad, z['mlp_grad_in'], 1e-4, 1e-7)
    for k, v in p.items():
        if v.requires_grad:
            close(v.grad, z['mlp_grad_' + k[len('readout_prob.'):]], 1e-4, 1e-6)
    close(bn['readout_prob.fc.1.running_mean'], z['mlp_after_running_mean1'], 1e-5, 1e-6)
    close(bn['readout_prob.fc.1.running_var'], z['mlp_after_running_var1'], 1e-5, 1e-6)
    close(bn['readout_prob.fc.5.running_mean'], z['mlp_after_running_mean5'], 1e-5, 1e-6)
    close(bn['readout_prob.fc.5.running_var'], z['mlp_after_running_var5'], 1e-5, 1e-6)


def test_func_loss_and_decoder():
    z = load('g3_ops')
    hf = torch.from_numpy(z['fl_hf']).requires_grad_(True)
    fl, dis = R.func_loss(hf, torch.from_numpy(z['fl_pairs']), torch.from_numpy(z['fl_tt']))
    close(dis, z['fl_dis'], 1e-5, 1e-6)
    close(fl, z['fl_loss'], 1e-5)
    fl.backward()
    close(hf.grad, z['fl_grad_hf'], 1e-4, 1e-7)
    s, t, ei = (torch.from_numpy(z[k]) for k in ('dec_s', 'dec_t', 'dec_edge_index'))
    close(R.decoder(s, t, ei), z['dec_sig'], 1e-6, 1e-7)
    close(R.decoder(s, t, ei, sigmoid=False), z['dec_raw'], 1e-6, 1e-6)


def test_vae_sampler_and_kl():
    z = load('g4_vae')
    p = R.params_from_npz(z)
    s = torch.from_numpy(z['s']).requires_grad_(True)
    t = torch.from_numpy(z['t']).requires_grad_(True)
    zs, zt, (smu, sls, tmu, tls) = R.vae_sample(p, s, t, torch.from_numpy(z['eps_s']), torch.from_numpy(z['eps_t']))
    close(zs, z['sample_s'], 1e-5, 1e-6)
    close(zt, z['sample_t'], 1e-5, 1e-6)
    skl, tkl = R.kl_term(smu, sls), R.kl_term(tmu, tls)
    close(skl, z['s_kl'], 1e-5)
    close(tkl, z['t_kl'], 1e-5)
    ((zs * torch.from_numpy(z['up_s'])).sum() + (zt * torch.from_numpy(z['up_t'])).sum()
     + float(z['kl_weight']) * (skl + tkl)).backward()
    close(s.grad, z['grad_s'], 1e-4, 1e-6)
    close(t.grad, z['grad_t'], 1e-4, 1e-6)
    for k, v in p.items():
        close(v.grad, z['grad_' + k], 1e-4, 1e-5)


def test_cfg1_trajectory_loose():
    """3 reference steps on BASELINE config 1 with dropout active: RNG streams differ, so only a loose
    trajectory check."""
    z = load('g5_cfg1')
    from deepgate import synthetic as syn
    b = syn.make_batch(1)
    batch = R.batch_from_arrays(lambda k: b[k])
    p = R.params_from_npz(z)
    bn = {k: v.clone() for k, v in p.items() if 'running_' in k}
    opt = torch.optim.Adam(R.trainable(p), lr=1e-4)
    plan = R.LevelPlan('aig', batch['edge_index'], batch['gate'], batch['forward_level'])
    torch.manual_seed(0)
    for step in range(3):
        opt.zero_grad()
        ls = R.run_batch(p, 'aig', batch, training=True, bn_state=bn, p_drop=0.2, plan=plan)
        R.weighted_loss(ls, [1.0, 4.0, 4.0]).backward()
        opt.step()
        got = [float(ls['recon_loss']), float(ls['prob_loss']), float(ls['func_loss'])]
        np.testing.assert_allclose(got, z['losses'][step], rtol=2e-2, atol=1e-2)


@pytest.mark.parametrize('name', ['g1_xmg', 'g2_aig'])
def test_fast_sweep_equals_plain_autograd_sweep(name):
    """The O(edges) level loop used for whole-workload CPU timing is the same function."""
    z = load(name)
    ctype, Rr, p, batch = _model_case(z)
    p2 = R.params_from_npz(z)
    outs = []
    for params, fast in ((p, False), (p2, True)):
        bn = {k: v.clone() for k, v in params.items() if 'running_' in k}
        ls = R.run_batch(params, ctype, batch, training=True, bn_state=bn, p_drop=0.0, s_rounds=Rr, t_rounds=Rr, fast=fast)
        R.weighted_loss(ls, z['meta_weights']).backward()
        outs.append(ls)
    close(outs[1]['hf'], outs[0]['hf'].detach().numpy(), 1e-6, 1e-6)
    for k in p:
        if p[k].requires_grad:
            ga = p[k].grad if p[k].grad is not None else torch.zeros_like(p[k])
            gb = p2[k].grad if p2[k].grad is not None else torch.zeros_like(p2[k])
            scale = max(1e-6, float(ga.abs().max()))
            np.testing.assert_allclose(gb.numpy(), ga.numpy(), rtol=1e-4, atol=1e-5 * scale + 1e-7, err_msg=k)


# ---- the literal variant (oracle/ref_cpu_literal.py: per-node subgraph scans, dense N x N mask, full-state level loop)
@pytest.mark.parametrize('name', ['g1_' + t for t in TYPES])
def test_literal_oracle_matches_reference_fixtures_and_the_vectorised_oracle(name):
    from oracle import ref_cpu_literal as L
    z = load(name)
    ctype, Rr, p, batch = _model_case(z)
    with torch.no_grad():
        hs, hf, s, t = L.model_forward(p, ctype, batch, Rr, Rr)
    close(hs, z['eval_hs'], 5e-5, 2e-5)
    close(hf, z['eval_hf'], 5e-5, 2e-5)
    bn = {k: v.clone() for k, v in p.items() if 'running_' in k}
    torch.manual_seed(0)
    ls = L.run_batch(p, ctype, batch, training=True, bn_state=bn, p_drop=0.0, s_rounds=Rr, t_rounds=Rr)
    close(ls['recon_loss'], z['train_recon'], 1e-5)          # the edge permutation does not move the mean
    close(ls['prob_loss'], z['train_prob_loss'], 1e-5)
    close(ls['func_loss'], z['train_func_loss'], 1e-5)
    R.weighted_loss(ls, z['meta_weights']).backward()
    for k, v in p.items():
        if v.requires_grad and v.grad is not None and ('grad_' + k) in z.files:
            ref = z['grad_' + k]
            np.testing.assert_allclose(v.grad.numpy(), ref, rtol=2e-4, atol=2e-4 * float(np.abs(ref).max()) + 2e-6, err_msg=k)


def test_literal_subgraph_and_edge_split_semantics():
    from oracle import ref_cpu_literal as L
    ei = torch.tensor([[0, 1, 2, 0, 3, 1], [3, 3, 4, 4, 5, 5]])
    sub = L.subgraph(torch.tensor([5, 3]), ei, dim=1)
    assert sub.tolist() == [[3, 1, 0, 1], [5, 5, 3, 3]]       # target order, original edge order inside a target
    torch.manual_seed(1)
    tp = L.general_train_test_split_edges(ei, 6)
    assert sorted(map(tuple, tp.t().tolist())) == sorted(map(tuple, ei.t().tolist()))
