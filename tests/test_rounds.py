"""Functional sweep with num_rounds > 1 (dg_ae_model_aig.py:70) and the stand-alone TFMlpAggr call (arch/tfmlp.py:31-46): both run on
HIP kernels only (level kernels' hidden-state variants, bf16x3 and exact fp32; csrc/attn_pool.hip).  CPU: the product refuses CPU
tensors.  GPU: two- and three-round models against the pinned oracle (bf16x3, exact-fp32 kernels, a batch with high fan-out
lists), the stand-alone aggregator against the reference's own fixture (g3_ops: TFMlpAggr + GRU from a NON-zero state, outputs and
every gradient) and against the oracle's restatement on random multigraphs."""
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


def test_standalone_aggregator_has_no_cpu_implementation():
    """The product path fails loudly on CPU tensors (no fallback): the CPU checker is oracle/ref_cpu.tf_mlp_aggr."""
    from deepgate import _hip
    from deepgate.arch.tfmlp import TFMlpAggr
    aggr = TFMlpAggr(32, 16)
    with pytest.raises(_hip.HipLibraryError):
        aggr(torch.randn(10, 32), torch.tensor([[0, 1, 2], [3, 3, 4]]))


def _check_against_oracle(model, sd, arrays, ctype, H, rounds, weights, tol):
    import deepgate
    from oracle import ref_cpu as R
    dev = torch.device('cuda:0')
    batch = deepgate.CircuitBatch.from_arrays(arrays, device=dev)
    tr = deepgate.Trainer(types.SimpleNamespace(model='DG_AE'), model, training_id='r2', save_dir='/tmp/mgv_test_exp', lr=1e-4,
                          rc_prob_func_weight=weights, device='cuda:0', batch_size=3, distributed=False)
    tr.optimizer.zero_grad()
    ls = tr.run_batch(batch)
    tr.weighted_loss(ls).backward()
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running_' not in k else v.clone()) for k, v in sd.items()}
    bn = {k: v.clone() for k, v in p.items() if 'running_' in k}
    ob = R.batch_from_arrays(lambda k: arrays[k])
    ols = R.run_batch(p, ctype, ob, training=True, bn_state=bn, p_drop=0.0, s_rounds=2, t_rounds=2, num_rounds=rounds)
    ols1 = R.run_batch({k: v.detach() for k, v in p.items()}, ctype, ob, training=True, bn_state={k: v.clone() for k, v in bn.items()},
                       p_drop=0.0, s_rounds=2, t_rounds=2, num_rounds=1)
    assert abs(float(ols['func_loss'].detach()) - float(ols1['func_loss'].detach())) > 1e-6      # the extra rounds do something
    R.weighted_loss(ols, weights).backward()
    for k in ('recon_loss', 'prob_loss', 'func_loss'):
        close(ls[k], ols[k].detach().numpy(), rtol=1e-4, msg=k)
    for k, q in model.named_parameters():
        ref = p[k].grad
        if q.grad is None or ref is None:
            assert (ref is None or float(ref.abs().max()) < 1e-5) and (q.grad is None or float(q.grad.abs().max()) < 1e-5), k
            continue
        g, ref = q.grad.detach().cpu().numpy(), ref.numpy()
        if 'attn_lin.weight' in k:
            g, ref = g[:, H:], ref[:, H:]
        scale = max(1e-6, float(np.abs(ref).max()))
        np.testing.assert_allclose(g, ref, rtol=2e-3, atol=tol * scale + 5e-6, err_msg='grad ' + k)
    return batch


def _model(ctype, H, rounds, seed=9):
    import deepgate
    torch.manual_seed(seed)
    enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=H, s_rounds=2, t_rounds=2, layernorm=True)
    model = getattr(deepgate, 'dg_ae_model_' + ctype).Model(struct_encoder=enc, num_rounds=rounds, dim_hidden=H)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    return model.to('cuda:0').train(), sd


@pytest.mark.gpu
@pytest.mark.parametrize('H,ctype,rounds,precision', [(64, 'aig', 2, 'x3'), (32, 'xmg', 2, 'x3'), (64, 'mig', 3, 'x3'),
                                                      (16, 'mig', 2, 'x3'), (64, 'xag', 2, 'f32')])
def test_two_round_model_against_the_oracle(H, ctype, rounds, precision):
    """(H = 16 and precision f32: the exact-fp32 level kernels' hidden-state variants, mgv_func_sweep_round_fwd / _bwd)"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from deepgate import ops, synthetic as syn
    old = ops.PRECISION
    ops.PRECISION = precision
    try:
        model, sd = _model(ctype, H, rounds)
        arrays = syn.collate([syn.make_graph(ctype, 150, 6, 700 + i, n_inputs=12) for i in range(3)])
        _check_against_oracle(model, sd, arrays, ctype, H, rounds, [1.0, 4.0, 4.0], 1e-3)
    finally:
        ops.PRECISION = old


@pytest.mark.gpu
def test_two_round_model_with_high_fanout_lists_against_the_oracle():
    """A primary input driving 700 gates and a level-1 gate driving 500: the sweep backward's heavy-list pre-passes, in rounds 1 and 2."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from deepgate import synthetic as syn
    ctype, H = 'aig', 64
    model, sd = _model(ctype, H, 2, seed=6)
    arrays = syn.collate([syn.make_graph(ctype, 1500, 12, 800 + i, n_inputs=24) for i in range(2)])
    rng = np.random.Generator(np.random.PCG64(4))
    n, ei, lv = arrays['num_nodes'], arrays['edge_index'], arrays['forward_level']
    have = set((ei[0] * n + ei[1]).tolist())
    dst = np.array([d for d in rng.choice(np.nonzero(lv > 0)[0], size=700, replace=False) if (5 * n + d) not in have])
    hub2 = int(np.nonzero(lv == 1)[0][3])
    dst2 = np.array([d for d in rng.choice(np.nonzero(lv >= 2)[0], size=500, replace=False) if (hub2 * n + d) not in have])
    arrays['edge_index'] = np.concatenate([ei, np.stack([np.full(len(dst), 5, dtype=ei.dtype), dst.astype(ei.dtype)]),
                                           np.stack([np.full(len(dst2), hub2, dtype=ei.dtype), dst2.astype(ei.dtype)])], axis=1)
    # (weights [1, 0, 4]: see test_hip_model.test_high_fanout_input_against_the_oracle)
    batch = _check_against_oracle(model, sd, arrays, ctype, H, 2, [1.0, 0.0, 4.0], 2e-3)
    plan = batch._mgv_plan
    assert plan.heavy_segments(True, inactive_only=True) is not None and plan.heavy_segments(True, active_by_level=True) is not None


@pytest.mark.gpu
def test_standalone_aggregator_on_hip_matches_the_reference_fixture():
    """TFMlpAggr.forward on a device tensor = csrc/attn_pool.hip + the linear kernels; outputs and every gradient of the
    reference's own fixture (aggregator + GRU from a non-zero state)."""
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
def test_standalone_aggregator_on_hip_against_the_oracle(H, N, E):
    """Random multigraphs (repeated edges, nodes without in-edges, one node with thousands of sources) against oracle.tf_mlp_aggr."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from deepgate.arch.tfmlp import TFMlpAggr
    from oracle import ref_cpu as R
    g = torch.Generator().manual_seed(H + N)
    dev = torch.device('cuda:0')
    ei = torch.stack([torch.randint(0, N, (E,), generator=g), torch.randint(N // 3, N, (E,), generator=g)])
    if E:
        ei[1, :E // 4] = N - 1                        # a heavy destination
    x = torch.randn(N, 2 * H, generator=g)
    up = torch.randn(N, H, generator=g)
    torch.manual_seed(5)
    hip = TFMlpAggr(2 * H, H)
    with torch.no_grad():
        hip.attn_lin.weight.mul_(4.0)                 # sharper softmax than the default init gives
    p = {'a.' + k: v.detach().clone().requires_grad_(True) for k, v in hip.state_dict().items()}
    hip = hip.to(dev)
    xr = x.clone().requires_grad_(True)
    xh = x.to(dev).requires_grad_(True)
    yr = R.tf_mlp_aggr(p, 'a', xr[ei[0]], xr[ei[1]], ei[1], N)
    yh = hip(xh, ei.to(dev))
    close(yh, yr, rtol=2e-4, atol=2e-5, msg='messages')
    (yr * up).sum().backward()
    (yh * up.to(dev)).sum().backward()
    close(xh.grad, xr.grad, rtol=1e-3, atol=1e-4, msg='grad x')
    for k, ph in hip.named_parameters():
        ref = p['a.' + k].grad
        if ph.grad is None:            # q side / biases inside the softmax: constant per segment, the oracle's gradient is rounding noise
            assert ref is None or float(ref.abs().max()) < 1e-4 * max(1.0, float(up.abs().max())), k
            continue
        g_, r_ = ph.grad.cpu(), ref
        if k == 'attn_lin.weight':
            g_, r_ = g_[:, H:], r_[:, H:]
        close(g_, r_, rtol=1e-3, atol=1e-4, msg='grad ' + k)


@pytest.mark.gpu
@pytest.mark.parametrize('ctype,T', [('xmg', 5), ('aig', 2)])
def test_grouped_round_linear_against_float64(ctype, T):
    """ops.RoundGhFn (csrc/linear_x3.hip grouped mode: W_hh[slot] h + b_hh[slot] for every updated gate in one launch over the sweep's
    tiles, partial tiles included; rounds >= 2 of dg_ae_model_aig.py:70,88-94) against float64 torch per gate type: gh, zero rows for the
    nodes no aggregator updates, and the gradients of h, W and b (bf16x3: 2e-4 of scale)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import deepgate
    from deepgate import ops, synthetic as syn
    if ops.PRECISION != 'x3':
        pytest.skip('bf16x3 mode only')
    dev = torch.device('cuda:0')
    H = 64
    g = [syn.make_graph(ctype, 150 + 37 * 45, 37, 11 + i, n_inputs=150) for i in range(2)]       # 45 nodes per level: partial tiles
    a = syn.collate(g)
    batch = deepgate.CircuitBatch.from_arrays(a, device=dev)
    gates = [gid for _, gid in getattr(deepgate, 'dg_ae_model_' + ctype).Model.GATES]
    assert len(gates) == T
    plan = deepgate.data.plan_of(batch, gates)
    N = plan.N
    torch.manual_seed(3 + T)
    h = torch.randn(N, H, device=dev, requires_grad=True)
    W = (torch.randn(T, 3 * H, H, device=dev) * 0.2).requires_grad_(True)
    b = (torch.randn(T, 3 * H, device=dev) * 0.1).requires_grad_(True)
    gout = torch.randn(N, 3 * H, device=dev)
    gh = ops.RoundGhFn.apply(plan, h, W, b)
    (gh * gout).sum().backward()
    h64, W64, b64 = (t.detach().double().cpu().requires_grad_(True) for t in (h, W, b))
    slot = plan.gslot.cpu().long()
    ref = torch.zeros(N, 3 * H, dtype=torch.float64)
    for s_ in range(T):
        idx = torch.nonzero(slot == s_).reshape(-1)
        ref = ref.index_add(0, idx, h64[idx] @ W64[s_].t() + b64[s_])
    (ref * gout.double().cpu()).sum().backward()
    idle = slot == 255
    assert int(idle.sum()) > 0 and float(gh.detach().cpu()[idle].abs().max()) == 0.0
    for name, got, want in (('gh', gh, ref), ('d h', h.grad, h64.grad), ('d W', W.grad, W64.grad), ('d b', b.grad, b64.grad)):
        want = want.detach()
        scale = float(want.abs().max())
        err = float((got.detach().cpu().double() - want).abs().max())
        assert scale > 0 and err <= 2e-4 * scale, (name, err, scale)
