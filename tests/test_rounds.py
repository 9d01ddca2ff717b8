"""Functional sweep with num_rounds > 1 (dg_ae_model_aig.py:70) and the stand-alone TFMlpAggr call (arch/tfmlp.py:31-46).
CPU: the composed level operator (the fallback of the exact-fp32 mode, H = 16 and high fan-out batches) against the reference's own
fixture (g3_ops: TFMlpAggr + GRU from a NON-zero state, outputs and every gradient), the composed round function against a plain
autograd restatement of the reference's level loop.  GPU: two- and three-round models against the pinned oracle THROUGH THE HIP
level kernels (mgv_func_sweep_round_*_x3: the composed round must not run)."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def close(a, b, rtol=2e-4, atol=2e-5, msg=''):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    scale = max(1e-6, float(np.abs(b).max()))
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * scale + 1e-7, err_msg=msg)


def test_standalone_aggregator_and_level_update_match_the_reference_fixture():
    from deepgate.arch.tfmlp import TFMlpAggr
    from deepgate._model_base import _level_update
    z = np.load(os.path.join(GOLDEN, 'g3_ops.npz'))
    H = z['lvl_hprev'].shape[1]
    aggr = TFMlpAggr(2 * H, H)
    aggr.load_state_dict({k[len('lvl_aggr_'):]: torch.tensor(z[k]) for k in z.files if k.startswith('lvl_aggr_')})
    gru = torch.nn.GRU(H, H)
    gru.load_state_dict({k[len('lvl_gru_'):]: torch.tensor(z[k]) for k in z.files if k.startswith('lvl_gru_')})
    ei = torch.tensor(z['lvl_edge_index'])
    nodes = torch.tensor(z['lvl_nodes'])
    ns = torch.tensor(z['lvl_node_state'], requires_grad=True)
    hprev = torch.tensor(z['lvl_hprev'], requires_grad=True)
    # the reference hands the aggregator the subgraph of edges INTO the level's nodes (utils/dag_utils.py:91-105)
    keep = torch.isin(ei[1], nodes)
    msg = aggr(ns, ei[:, keep])
    close(msg[nodes], z['lvl_msg'], msg='stand-alone TFMlpAggr.forward')
    others = torch.ones(ns.shape[0], dtype=torch.bool)
    others[nodes] = False
    assert float(msg[others].detach().abs().max()) == 0.0
    # level update from a non-zero state: gathered-row form used by the multi-round path
    pos = torch.full((ns.shape[0],), -1, dtype=torch.long)
    pos[nodes] = torch.arange(nodes.numel())
    src, seg = ei[0, keep], pos[ei[1, keep]]
    hnew = _level_update(aggr, gru, ns[src, :H], ns[src, H:], seg, nodes.numel(), hprev[nodes])
    close(hnew, z['lvl_hnew'], msg='hnew')
    (hnew * torch.tensor(z['lvl_up'])).sum().backward()
    close(ns.grad, z['lvl_grad_node_state'], rtol=1e-3, atol=1e-4, msg='grad node_state')
    close(hprev.grad, z['lvl_grad_hprev'], rtol=1e-3, atol=1e-4, msg='grad hprev')
    for k, p in aggr.named_parameters():
        ref = z['lvl_grad_aggr_' + k]
        if p.grad is None:             # q side / biases inside the softmax: constant per segment, reference gradient is rounding noise
            assert float(np.abs(ref).max()) < 1e-5, k
            continue
        g = p.grad.numpy()
        if k == 'attn_lin.weight':
            g, ref = g[:, H:], ref[:, H:]
        close(g, ref, rtol=1e-3, atol=1e-4, msg='grad aggr ' + k)
    for k, p in gru.named_parameters():
        close(p.grad, z['lvl_grad_gru_' + k], rtol=1e-3, atol=1e-4, msg='grad gru ' + k)


def _plain_round(plan, mods, hs, hf):
    """The reference's level loop with plain autograd (index_put per group)."""
    from deepgate._model_base import _level_update
    for nodes, slot, src, seg in plan.level_groups():
        aggr, gru = mods[slot]
        hn = _level_update(aggr, gru, hs[src], hf[src], seg, nodes.numel(), hf[nodes])
        hf = hf.index_put((nodes,), hn)
    return hf


@pytest.mark.parametrize('ctype', ['aig', 'xmg'])
def test_round_function_equals_plain_autograd(ctype):
    import deepgate
    from deepgate import synthetic as syn
    from deepgate._model_base import ExtraRoundFn, _round_params
    from deepgate.graph_plan import GraphPlan
    H = 16
    torch.manual_seed(3)
    mod = getattr(deepgate, 'dg_ae_model_' + ctype)
    gate_ids = [g for _, g in mod.Model.GATES]
    arrays = syn.collate([syn.make_graph(ctype, 122, 7, 40 + i, n_inputs=10) for i in range(2)])
    plan = GraphPlan(torch.from_numpy(arrays['edge_index']), arrays['num_nodes'])
    plan.set_levels(torch.from_numpy(arrays['gate']), torch.from_numpy(arrays['forward_level']), gate_ids)
    mods = [(deepgate.arch.tfmlp.TFMlpAggr(2 * H, H), torch.nn.GRU(H, H)) for _ in gate_ids]
    N = arrays['num_nodes']
    up = torch.randn(N, H)
    res = []
    for fn in ('fn', 'plain'):
        hs = torch.randn(N, H, generator=torch.Generator().manual_seed(1)).requires_grad_(True)
        hf0 = torch.randn(N, H, generator=torch.Generator().manual_seed(2)).requires_grad_(True)
        for a, g in mods:
            a.zero_grad(); g.zero_grad()
        if fn == 'fn':
            out = ExtraRoundFn.apply(plan, mods, hs, hf0, *[p for a, g in mods for p in _round_params(a, g)])
            out = ExtraRoundFn.apply(plan, mods, hs, out, *[p for a, g in mods for p in _round_params(a, g)])     # rounds chain
        else:
            out = _plain_round(plan, mods, hs, _plain_round(plan, mods, hs, hf0))
        (out * up).sum().backward()
        res.append((out.detach(), hs.grad.clone(), hf0.grad.clone(),
                    [None if p.grad is None else p.grad.clone() for a, g in mods for p in _round_params(a, g)]))
    close(res[0][0], res[1][0], rtol=1e-5, atol=1e-6, msg='hf')
    close(res[0][1], res[1][1], rtol=1e-4, atol=1e-5, msg='grad hs')
    close(res[0][2], res[1][2], rtol=1e-4, atol=1e-5, msg='grad hf_in')
    for i, (a, b) in enumerate(zip(res[0][3], res[1][3])):
        assert (a is None) == (b is None), i
        if a is not None:
            close(a, b, rtol=1e-4, atol=1e-5, msg='param %d' % i)


@pytest.mark.gpu
@pytest.mark.parametrize('H,ctype,rounds', [(64, 'aig', 2), (32, 'xmg', 2), (64, 'mig', 3)])
def test_two_round_model_against_the_oracle(H, ctype, rounds, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import deepgate
    from deepgate import _model_base, synthetic as syn
    from oracle import ref_cpu as R

    def composed_round_must_not_run(*a, **k):
        raise AssertionError('rounds >= 2 left the HIP level kernels')
    monkeypatch.setattr(_model_base.ExtraRoundFn, 'apply', composed_round_must_not_run)
    dev = torch.device('cuda:0')
    torch.manual_seed(9)
    enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=H, s_rounds=2, t_rounds=2, layernorm=True)
    model = getattr(deepgate, 'dg_ae_model_' + ctype).Model(struct_encoder=enc, num_rounds=rounds, dim_hidden=H)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(dev).train()
    arrays = syn.collate([syn.make_graph(ctype, 150, 6, 700 + i, n_inputs=12) for i in range(3)])
    batch = deepgate.CircuitBatch.from_arrays(arrays, device=dev)
    tr = deepgate.Trainer(types.SimpleNamespace(model='DG_AE'), model, training_id='r2', save_dir='/tmp/mgv_test_exp', lr=1e-4,
                          rc_prob_func_weight=[1.0, 4.0, 4.0], device='cuda:0', batch_size=3, distributed=False)
    tr.optimizer.zero_grad()
    ls = tr.run_batch(batch)
    tr.weighted_loss(ls).backward()
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running_' not in k else v.clone()) for k, v in sd.items()}
    bn = {k: v.clone() for k, v in p.items() if 'running_' in k}
    ob = R.batch_from_arrays(lambda k: arrays[k])
    ols = R.run_batch(p, ctype, ob, training=True, bn_state=bn, p_drop=0.0, s_rounds=2, t_rounds=2, num_rounds=rounds)
    ols1 = R.run_batch({k: v.detach() for k, v in p.items()}, ctype, ob, training=True, bn_state={k: v.clone() for k, v in bn.items()},
                       p_drop=0.0, s_rounds=2, t_rounds=2, num_rounds=1)
    assert abs(float(ols['prob_loss'].detach()) - float(ols1['prob_loss'].detach())) > 1e-6      # the second round does something
    R.weighted_loss(ols, [1.0, 4.0, 4.0]).backward()
    for k in ('recon_loss', 'prob_loss', 'func_loss'):
        close(ls[k], ols[k].detach().numpy(), rtol=1e-4, msg=k)
    for k, q in model.named_parameters():
        ref = p[k].grad
        if q.grad is None:
            assert ref is None or float(ref.abs().max()) < 1e-5, k
            continue
        if ref is None:
            assert float(q.grad.abs().max()) == 0.0, k
            continue
        g, ref = q.grad.detach().cpu().numpy(), ref.numpy()
        if 'attn_lin.weight' in k:
            g, ref = g[:, H:], ref[:, H:]
        scale = max(1e-6, float(np.abs(ref).max()))
        np.testing.assert_allclose(g, ref, rtol=2e-3, atol=1e-3 * scale + 5e-6, err_msg='grad ' + k)


@pytest.mark.gpu
def test_standalone_aggregator_on_hip_matches_the_reference_fixture():
    """TFMlpAggr.forward on a device tensor = csrc/attn_pool.hip + the linear kernels; outputs and every gradient of the
    reference's own fixture (aggregator + GRU from a non-zero state).  The composed PyTorch form must not run."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from deepgate.arch.tfmlp import TFMlpAggr
    z = np.load(os.path.join(GOLDEN, 'g3_ops.npz'))
    H = z['lvl_hprev'].shape[1]
    dev = torch.device('cuda:0')
    aggr = TFMlpAggr(2 * H, H)
    aggr.load_state_dict({k[len('lvl_aggr_'):]: torch.tensor(z[k]) for k in z.files if k.startswith('lvl_aggr_')})
    gru = torch.nn.GRU(H, H)
    gru.load_state_dict({k[len('lvl_gru_'):]: torch.tensor(z[k]) for k in z.files if k.startswith('lvl_gru_')})
    aggr, gru = aggr.to(dev), gru.to(dev)
    aggr.attend = None                       # the composed form is the CPU path
    ei = torch.tensor(z['lvl_edge_index'])
    nodes = torch.tensor(z['lvl_nodes'])
    keep = torch.isin(ei[1], nodes)
    ns = torch.tensor(z['lvl_node_state'], device=dev, requires_grad=True)
    hprev = torch.tensor(z['lvl_hprev'], device=dev, requires_grad=True)
    msg = aggr(ns, ei[:, keep].to(dev))
    nd = nodes.to(dev)
    close(msg[nd], z['lvl_msg'], msg='stand-alone TFMlpAggr.forward (HIP)')
    others = torch.ones(ns.shape[0], dtype=torch.bool, device=dev)
    others[nd] = False
    assert float(msg[others].detach().abs().max()) == 0.0
    hnew = gru(msg[nd].unsqueeze(0), hprev[nd].unsqueeze(0))[1][0]
    close(hnew, z['lvl_hnew'], msg='hnew')
    (hnew * torch.tensor(z['lvl_up'], device=dev)).sum().backward()
    close(ns.grad, z['lvl_grad_node_state'], rtol=1e-3, atol=1e-4, msg='grad node_state')
    close(hprev.grad, z['lvl_grad_hprev'], rtol=1e-3, atol=1e-4, msg='grad hprev')
    for k, p in aggr.named_parameters():
        ref = z['lvl_grad_aggr_' + k]
        if p.grad is None:
            assert float(np.abs(ref).max()) < 1e-5, k
            continue
        g = p.grad.cpu().numpy()
        if k == 'attn_lin.weight':
            g, ref = g[:, H:], ref[:, H:]
        close(g, ref, rtol=1e-3, atol=1e-4, msg='grad aggr ' + k)
    for k, p in gru.named_parameters():
        close(p.grad, z['lvl_grad_gru_' + k], rtol=1e-3, atol=1e-4, msg='grad gru ' + k)


@pytest.mark.gpu
@pytest.mark.parametrize('H,N,E', [(16, 300, 900), (32, 5000, 20000), (64, 777, 12000), (32, 64, 0)])
def test_standalone_aggregator_on_hip_against_the_composed_form(H, N, E):
    """Random multigraphs (repeated edges, nodes without in-edges, one node with thousands of sources) against the CPU form."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from deepgate.arch.tfmlp import TFMlpAggr
    g = torch.Generator().manual_seed(H + N)
    dev = torch.device('cuda:0')
    ei = torch.stack([torch.randint(0, N, (E,), generator=g), torch.randint(N // 3, N, (E,), generator=g)])
    if E:
        ei[1, :E // 4] = N - 1                        # a heavy destination
    x = torch.randn(N, 2 * H, generator=g)
    up = torch.randn(N, H, generator=g)
    torch.manual_seed(5)
    ref = TFMlpAggr(2 * H, H)
    with torch.no_grad():
        ref.attn_lin.weight.mul_(4.0)                 # sharper softmax than the default init gives
    hip = TFMlpAggr(2 * H, H)
    hip.load_state_dict(ref.state_dict())
    hip = hip.to(dev)
    xr = x.clone().requires_grad_(True)
    xh = x.to(dev).requires_grad_(True)
    yr = ref(xr, ei)
    yh = hip(xh, ei.to(dev))
    close(yh, yr, rtol=2e-4, atol=2e-5, msg='messages')
    (yr * up).sum().backward()
    (yh * up.to(dev)).sum().backward()
    close(xh.grad, xr.grad, rtol=1e-3, atol=1e-4, msg='grad x')
    for (k, pr), (_, ph) in zip(ref.named_parameters(), hip.named_parameters()):
        assert (pr.grad is None) == (ph.grad is None), k
        if pr.grad is not None:
            close(ph.grad, pr.grad, rtol=1e-3, atol=1e-4, msg='grad ' + k)
