"""GPU tests of the user-visible surface around the hot path: checkpoints (Trainer.save / load / resume, the reference's own
.pth written by DG_VAE/deepgate/trainer.py:105-111 and read back by :113-129 + utils/model_utils.py:3-66), optimiser state
interchange with torch.optim.Adam, the train.py entry (train.py:45-104) and the feature-extraction example
(examples/feature_extract_bench.py:13-25).  Fixtures: tests/golden/g7_ckpt* (make_golden.py, reference run)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN, PKG_PARENT

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def _model(ctype, H, R, seed):
    import deepgate
    torch.manual_seed(seed)
    enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=H, s_rounds=R, t_rounds=R, layernorm=True)
    model = getattr(deepgate, 'dg_ae_model_' + ctype).Model(struct_encoder=enc, dim_hidden=H)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model


def _trainer(model, save_dir, tid='t'):
    import deepgate
    return deepgate.Trainer(types.SimpleNamespace(model='DG_AE'), model, training_id=tid, save_dir=str(save_dir), lr=1e-4,
                            rc_prob_func_weight=[1.0, 4.0, 4.0], device='cuda:0', batch_size=1, distributed=False)


def _batch(dev, n=3, nodes=256):
    import deepgate
    from deepgate import synthetic as syn
    graphs = [syn.make_graph('aig', nodes, 8, 4400 + i, n_inputs=16) for i in range(n)]
    return deepgate.CircuitBatch.from_arrays(syn.collate(graphs), device=dev)


def _snapshot(tr):
    f = tr.optimizer.flat_buffers()
    return {'p': f['param'].clone(), 'm': f['m'].clone(), 'v': f['v'].clone(), 'step': tr.optimizer._step,
            'lr': tr.optimizer.param_groups[0]['lr'], 'bn': {k: v.clone() for k, v in tr.model.state_dict().items() if 'running_' in k or 'num_batches' in k}}


def test_save_load_resume_restore_the_exact_state_and_continue_like_an_uninterrupted_run(tmp_path):
    dev = _dev()
    batch = _batch(dev)
    a = _trainer(_model('aig', 64, 2, 5), tmp_path, 'a')
    a.model.train()
    for _ in range(2):
        a.train_step(batch)
    a.model_epoch = 7
    a.set_training_args(lr=3e-4)
    ckpt = os.path.join(a.log_dir, 'model_last.pth')
    a.save(ckpt)
    saved = _snapshot(a)
    a.train_step(batch)
    after3 = _snapshot(a)

    for how in ('load', 'resume'):
        b = _trainer(_model('aig', 64, 2, 99), tmp_path, 'a')          # other initial weights: everything must come from the file
        b.model.train()
        if how == 'load':
            b.load(ckpt)
        else:
            assert b.resume() is True
        assert b.model_epoch == 7
        got = _snapshot(b)                     # flattens: the deferred optimiser state is applied here
        assert got['lr'] == saved['lr'] == 3e-4 and got['step'] == saved['step'] == 2
        for k in ('p', 'm', 'v'):
            assert torch.equal(got[k], saved[k]), (how, k)      # restored bit for bit
        for k, v in saved['bn'].items():
            assert torch.equal(got['bn'][k], v), (how, k)
        b.train_step(batch)
        nxt = _snapshot(b)
        assert nxt['step'] == 3
        # step 3 equals the uninterrupted run's; a few kernels still sum with float atomics (order varies run to run), so
        # entries whose gradient is rounding noise may move by up to +-lr: compare where the first moment is not noise
        live = after3['m'].abs() > 1e-6
        d = (nxt['p'] - after3['p']).abs()
        assert float(d[live].max()) <= 2e-6, (how, float(d[live].max()))
        assert float(d.max()) <= 2.1 * 3e-4
        assert float((nxt['m'] - after3['m']).abs().max()) <= 1e-5 * float(after3['m'].abs().max()) + 1e-9


def test_checkpoint_written_by_the_reference_trainer_loads_and_continues_like_the_reference(tmp_path):
    """g7: the reference's Trainer stepped once, saved, a fresh reference Trainer loaded the file and stepped again.
    The same file through our Trainer.load must give the same restored weights and the same next step."""
    dev = _dev()
    import deepgate
    z = np.load(os.path.join(GOLDEN, 'g7_ckpt.npz'))
    path = os.path.join(GOLDEN, 'g7_ckpt_ref_aig.pth')
    model = _model(str(z['meta_type']), int(z['meta_H']), int(z['meta_R']), seed=1234)
    tr = _trainer(model, tmp_path)
    tr.load(path)
    assert tr.model_epoch == int(z['meta_epoch']) and abs(tr.lr - 1e-4) < 1e-12
    sd = model.state_dict()
    for k in sd:
        np.testing.assert_array_equal(sd[k].cpu().numpy(), z['saved_' + k], err_msg=k)
    batch = deepgate.CircuitBatch.from_arrays({k[3:]: z[k] for k in z.files if k.startswith('in_')}, device=dev)
    model.train()
    ls = tr.train_step(batch)
    got = [float(ls[k].detach()) for k in ('recon_loss', 'prob_loss', 'func_loss')]
    np.testing.assert_allclose(got, z['losses_step2'], rtol=1e-4, atol=1e-6)
    assert tr.optimizer._step == int(z['adam2_step']) == 2
    f = tr.optimizer.flat_buffers()
    for (k, p), (_, _, off) in zip([(k, p) for k, p in model.named_parameters() if p.requires_grad], tr.optimizer._indexed()):
        n = p.numel()
        if 'adam2_exp_avg_' + k not in z.files:
            continue
        m_ref = z['adam2_exp_avg_' + k].reshape(-1)
        live = np.abs(m_ref) > 1e-6
        m_got = f['m'][off:off + n].cpu().numpy()
        np.testing.assert_allclose(m_got[live], m_ref[live], rtol=2e-3, atol=1e-7, err_msg='exp_avg ' + k)
        np.testing.assert_allclose(p.detach().cpu().numpy().reshape(-1)[live], z['after2_' + k].reshape(-1)[live], rtol=1e-5, atol=3e-6,
                                   err_msg='parameter after the resumed step: ' + k)


def test_flat_adam_speaks_torch_adam_state_dicts():
    dev = _dev()
    from deepgate.optim import FlatAdam
    torch.manual_seed(3)
    ps_a = [torch.nn.Parameter(torch.randn(7, 5, device=dev)), torch.nn.Parameter(torch.randn(11, device=dev)),
            torch.nn.Parameter(torch.randn(3, device=dev), requires_grad=False), torch.nn.Parameter(torch.randn(4, 4, device=dev))]
    ps_b = [torch.nn.Parameter(p.detach().clone(), requires_grad=p.requires_grad) for p in ps_a]
    grads = [[torch.randn_like(p) for p in ps_a] for _ in range(3)]
    ref = torch.optim.Adam(ps_a, lr=1e-2)
    for g in grads[:2]:
        for p, gg in zip(ps_a, g):
            p.grad = gg.clone() if p.requires_grad else None
        ref.step()
    mine = FlatAdam(ps_b, lr=1.0)
    for p, q in zip(ps_b, ps_a):
        p.data.copy_(q.data)
    mine.load_state_dict(ref.state_dict())          # before the first flatten: deferred path; index 2 is the frozen parameter
    for opt, ps in ((ref, ps_a), (mine, ps_b)):
        for p, gg in zip(ps, grads[2]):
            p.grad = gg.clone() if p.requires_grad else None
        opt.step()
    assert mine.param_groups[0]['lr'] == 1e-2 and mine._step == 3
    for p, q in zip(ps_b, ps_a):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=1e-6, atol=1e-7)
    back = mine.state_dict()
    assert sorted(back['state']) == [0, 1, 3] and back['param_groups'][0]['params'] == [0, 1, 2, 3]
    for i in (0, 1, 3):
        np.testing.assert_allclose(back['state'][i]['exp_avg'].cpu().numpy(), ref.state_dict()['state'][i]['exp_avg'].cpu().numpy(), rtol=2e-5, atol=1e-7)
    fresh = torch.optim.Adam([torch.nn.Parameter(p.detach().clone(), requires_grad=p.requires_grad) for p in ps_b], lr=1.0)
    fresh.load_state_dict(back)                     # and torch.optim.Adam reads what FlatAdam writes


def test_train_entry_and_feature_extraction(tmp_path, monkeypatch):
    """`python train.py --model DG_AE --type aig --synthetic ...` through its three stages (train.py:81-101), then the
    example script embeds graphs with the checkpoint it wrote and agrees with a direct Model.forward."""
    dev = _dev()
    monkeypatch.syspath_prepend(PKG_PARENT)
    import importlib
    train = importlib.import_module('train')
    train.main(['--exp_id', 'e', '--model', 'DG_AE', '--type', 'aig', '--layernorm', '--batch_size', '2', '--synthetic', '6',
                '--synthetic_nodes', '256', '--synthetic_levels', '8', '--stage_epochs', '1', '1', '1', '--s_rounds', '2', '--t_rounds', '2',
                '--save_dir', str(tmp_path)])
    ckpt = tmp_path / 'e' / 'stage_3.pth'
    assert ckpt.exists() and (tmp_path / 'e' / 'model_last.pth').exists()
    cp = torch.load(ckpt, map_location='cpu')
    assert set(cp) == {'epoch', 'state_dict', 'optimizer'} and cp['epoch'] == 3
    assert all(torch.isfinite(v).all() for v in cp['state_dict'].values() if v.is_floating_point())
    log = [f for f in os.listdir(tmp_path / 'e') if f.startswith('log-')]
    assert log and 'train| Epoch' in open(tmp_path / 'e' / log[0]).read()

    sys.path.insert(0, os.path.join(PKG_PARENT, 'examples'))
    fe = importlib.import_module('feature_extract')
    out = tmp_path / 'emb.npz'
    fe.main(['--type', 'aig', '--synthetic', '3', '--checkpoint', str(ckpt), '--rounds', '2', '--batch_size', '2', '--out', str(out)])
    emb = np.load(out)
    assert sorted(emb.files) == sorted('graph%d/%s' % (i, k) for i in range(3) for k in ('hs', 'hf'))
    import deepgate
    from deepgate import synthetic as syn
    model = _model('aig', 64, 2, 0).to(dev)
    model.load(str(ckpt))
    model.eval()
    g = syn.make_graph('aig', 1024, 30, 900, n_inputs=64)
    with torch.no_grad():
        hs, hf = model(deepgate.CircuitBatch.from_arrays(syn.collate([g]), device=dev))
    np.testing.assert_allclose(emb['graph0/hs'], hs.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(emb['graph0/hf'], hf.cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_two_identical_steps_give_bit_identical_gradients():
    """Same parameters, same batch, same negatives, twice: every parameter gradient is bit-identical (no float atomics on the
    H = 64 path: per-workgroup slabs and fixed-order reductions)."""
    dev = _dev()
    import types
    import deepgate
    from deepgate import synthetic as syn
    torch.manual_seed(11)
    enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=64, s_rounds=2, t_rounds=2, layernorm=True)
    model = deepgate.dg_ae_model_aig.Model(struct_encoder=enc, dim_hidden=64).to(dev).train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    arrays = syn.collate([syn.make_graph('aig', 4096, 30, 900 + i, n_inputs=256) for i in range(4)])
    batch = deepgate.CircuitBatch.from_arrays(arrays, device=dev)
    N = arrays['num_nodes']
    g = torch.Generator().manual_seed(3)
    E = arrays['edge_index'].shape[1]
    batch.neg_edge_index = torch.stack([torch.randint(0, N, (E + N,), generator=g), torch.randint(0, N, (E + N,), generator=g)]).to(dev)
    tr = deepgate.Trainer(types.SimpleNamespace(model='DG_AE'), model, training_id='det', save_dir='/tmp/mgv_test_exp', lr=1e-4,
                          rc_prob_func_weight=[1.0, 4.0, 4.0], device='cuda:0', batch_size=4, distributed=False)
    grads = []
    for _ in range(2):
        tr.optimizer.zero_grad()
        ls = tr.run_batch(batch, want_pred=False)
        tr.weighted_loss(ls).backward()
        torch.cuda.synchronize()
        grads.append({k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    diff = [k for k in grads[0] if not torch.equal(grads[0][k], grads[1][k])]
    assert not diff, diff
