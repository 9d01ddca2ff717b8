"""GPU parity: structural-encoder kernels (C ABI -> ops -> modules) against the golden vectors and
the oracle.  Tolerances: fp32 arithmetic with a different (but fixed) summation order than the CPU
reference -> 2e-4 relative on activations / gradients, stated per assert."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def close(a, b, rtol=2e-4, atol=2e-5, msg=''):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    scale = max(1.0, float(np.abs(b).max()))
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol * scale, err_msg=msg)


def test_library_loads_and_reports_abi():
    _dev()
    from deepgate import _hip
    lib = _hip.load()
    assert lib.mgv_abi_version() >= 1


def test_half_rounds_forward_backward_vs_reference_fixture():
    dev = _dev()
    from deepgate import ops
    from deepgate.graph_plan import GraphPlan
    z = load('g3_ops')
    H = 64
    raw = {k[len('enc_param_'):]: torch.tensor(z[k], device=dev, requires_grad=True) for k in z.files if k.startswith('enc_param_')}
    ei = torch.tensor(z['enc_edge_index'], device=dev)
    x = torch.tensor(z['enc_x'], device=dev)
    N = x.shape[0]
    plan = GraphPlan(ei, N)
    rows = torch.eye(6, device=dev)
    xcls = x.argmax(1).to(torch.uint8)

    def composed(aggr, gru):
        w_ih = raw[gru + '.weight_ih_l0']
        Wc = w_ih[:, :H] @ raw[aggr + '.msg.weight']
        bc = w_ih[:, :H] @ raw[aggr + '.msg.bias']
        xtab = rows @ w_ih[:, H:].t() + raw[gru + '.bias_ih_l0']
        return [xtab, Wc, bc, raw[gru + '.weight_hh_l0'], raw[gru + '.bias_hh_l0']]

    cf, cr = composed('aggr', 'update'), composed('aggr_r', 'update_r')
    df = [t.detach().contiguous() for t in cf]
    dr = [t.detach().contiguous() for t in cr]
    lw, lb = raw['ln.weight'].detach(), raw['ln.bias'].detach()
    h0 = torch.tensor(z['enc_h0'], device=dev)
    h1 = ops.struct_stage_fwd(h0, plan.in_ptr, plan.in_src, xcls, *df, lw, lb)
    close(h1, z['enc_h1'], msg='forward half round')
    h2 = ops.struct_stage_fwd(h1, plan.out_ptr, plan.out_dst, xcls, *dr, lw, lb)
    close(h2, z['enc_h2'], msg='reversed half round')

    def accs(d):
        return {'dxtab': torch.zeros_like(d[0]), 'dWc': torch.zeros_like(d[1]), 'dbc': torch.zeros_like(d[2]),
                'dWhh': torch.zeros_like(d[3]), 'dbhh': torch.zeros_like(d[4])}
    gf, gr = accs(df), accs(dr)
    dlw, dlb = torch.zeros_like(lw), torch.zeros_like(lb)
    up = torch.tensor(z['enc_up'], device=dev)
    gd, ga = ops.struct_stage_bwd(h1, plan.out_ptr, plan.out_dst, xcls, *dr, lw, lb, up, None,
                                  dict(gr, dln_w=dlw, dln_b=dlb))
    gd0, ga0 = ops.struct_stage_bwd(h0, plan.in_ptr, plan.in_src, xcls, *df, lw, lb, gd, ga,
                                    dict(gf, dln_w=dlw, dln_b=dlb))
    # dL/dh0 = direct part + scatter of the aggregate gradient back along the in-edges
    g_h0 = gd0.clone()
    g_h0.index_add_(0, ei[0], ga0[ei[1]])
    close(g_h0, z['enc_grad_h0'], msg='grad h0')
    close(dlw, z['enc_grad_ln.weight'], msg='grad ln.weight')
    close(dlb, z['enc_grad_ln.bias'], msg='grad ln.bias')
    # raw-parameter gradients through the (torch-side) weight composition
    torch.autograd.backward(cf + cr, [gf['dxtab'], gf['dWc'], gf['dbc'], gf['dWhh'], gf['dbhh'],
                                      gr['dxtab'], gr['dWc'], gr['dbc'], gr['dWhh'], gr['dbhh']])
    for k, v in raw.items():
        if k.startswith('ln.'):
            continue
        close(v.grad, z['enc_grad_' + k], msg='grad ' + k)


@pytest.mark.parametrize('name', ['g1_aig', 'g1_xmg', 'g2_aig'])
def test_direct_encoder_module_vs_reference_outputs(name):
    dev = _dev()
    import deepgate
    from deepgate.digae_layer import DirectMultiGCNEncoder
    z = load(name)
    H, R = int(z['meta_H']), int(z['meta_R'])
    ctype = str(z['meta_type'])
    prefix = {'aig': 'struct_encoder'}.get(ctype, ctype + '_struct_encoder') + '.'
    enc = DirectMultiGCNEncoder(dim_feature=6, dim_hidden=H, s_rounds=R, t_rounds=R, layernorm=True).to(dev)
    sd = {k[len('param_') + len(prefix):]: torch.tensor(z[k]) for k in z.files if k.startswith('param_' + prefix)}
    enc.load_state_dict(sd, strict=True)
    x = torch.tensor(z['in_x'], device=dev)
    one_hot = torch.nn.functional.one_hot(x[:, 1].long(), num_classes=6)
    ei = torch.tensor(z['in_edge_index'], device=dev)
    s, t = enc(one_hot, one_hot, ei)
    close(s, z['eval_s'], msg='s')
    close(t, z['eval_t'], msg='t')
    # gradients of a random projection against the oracle's autograd
    from oracle import ref_cpu as R_
    up_s = torch.randn(s.shape, generator=torch.Generator().manual_seed(3))
    up_t = torch.randn(t.shape, generator=torch.Generator().manual_seed(4))
    ((s * up_s.to(dev)).sum() + (t * up_t.to(dev)).sum()).backward()
    p = R_.params_from_npz(z)
    so, to = R_.struct_encoder(p, prefix[:-1], one_hot.cpu(), ei.cpu(), R, R, True)
    ((so * up_s).sum() + (to * up_t).sum()).backward()
    for k, v in enc.named_parameters():
        ref = p[prefix + k].grad
        close(v.grad, ref, rtol=5e-4, atol=5e-5, msg='grad ' + k)


@pytest.mark.parametrize('H,F_', [(64, 6), (16, 3)])
def test_general_node_features_against_the_oracle(H, F_):
    """digae_layer.py:257-277 takes any x [N, F]; the reference Models only feed one-hot rows.  With random float features (every row
    distinct) the encoder forms the GRU's feature term per node (MultiGCNEncoder._forward_rows, ops.StructEncoderRowsFn on the
    exact-fp32 stage kernels): s, t and every parameter gradient — and the gradient of x itself — against oracle/ref_cpu.py."""
    dev = _dev()
    import deepgate
    from deepgate import synthetic as syn
    from oracle import ref_cpu as R
    arrays = syn.collate([syn.make_graph('xmg', 60 + 12 * 54, 12, 40 + i, n_inputs=60) for i in range(2)])
    n = arrays['num_nodes']
    rng = np.random.Generator(np.random.PCG64(3))
    x_np = rng.standard_normal((n, F_)).astype(np.float32)
    ei_np = arrays['edge_index']
    torch.manual_seed(7)
    enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=F_, dim_hidden=H, s_rounds=2, t_rounds=1, layernorm=True)
    sd = {k: v.clone() for k, v in enc.state_dict().items()}
    enc = enc.to(dev)
    x = torch.from_numpy(x_np).to(dev).requires_grad_(True)
    ei = torch.from_numpy(ei_np).to(dev)
    assert deepgate.digae_layer.feature_classes(x.detach()) is None          # more distinct rows than the class table holds
    s, t = enc(x, x, ei)
    gs = torch.from_numpy(rng.standard_normal((n, H)).astype(np.float32))
    gt = torch.from_numpy(rng.standard_normal((n, H)).astype(np.float32))
    ((s * gs.to(dev)).sum() + (t * gt.to(dev)).sum()).backward()
    p = {'enc.' + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xo = torch.from_numpy(x_np).requires_grad_(True)
    so, to = R.struct_encoder(p, 'enc', xo, torch.from_numpy(ei_np), 2, 1, layernorm=True)
    ((so * gs).sum() + (to * gt).sum()).backward()
    for nm, a, b in (('s', s, so), ('t', t, to)):
        assert float((a.detach().cpu() - b.detach()).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max())), nm
    named = dict(enc.named_parameters())
    for k, v in p.items():
        g, ref = named[k[4:]].grad, v.grad
        scale = float(ref.abs().max())
        assert g is not None and scale > 0, k
        assert float((g.cpu() - ref).abs().max()) <= 1e-3 * scale, (k, float((g.cpu() - ref).abs().max()) / scale)
    assert float((x.grad.cpu() - xo.grad).abs().max()) <= 1e-3 * float(xo.grad.abs().max())
