"""Diagnostic: per-parameter gradient deviation of one degenerate case (tests/test_hip_degenerate.py) against the oracle."""
import sys, os, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'multi-gate-vae_amd'), os.path.join(ROOT, 'tests')]
import test_hip_degenerate as T
import deepgate
from deepgate import synthetic as syn
from oracle import ref_cpu as R

def run(ctype, graphs, weights, rounds, dtype=torch.float32):
    H = 64
    torch.manual_seed(3)
    enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=H, s_rounds=rounds, t_rounds=rounds, layernorm=True)
    model = deepgate.dg_ae_model_aig.Model(struct_encoder=enc, dim_hidden=H)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout): m.p = 0.0
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to('cuda:0').train()
    arrays = syn.collate(graphs)
    batch = deepgate.CircuitBatch.from_arrays(arrays, device='cuda:0')
    tr = deepgate.Trainer(types.SimpleNamespace(model='DG_AE'), model, training_id='deg', save_dir='/tmp/mgv_test_exp', lr=1e-4,
                          rc_prob_func_weight=list(weights), device='cuda:0', batch_size=len(graphs), distributed=False)
    tr.optimizer.zero_grad()
    ls = tr.run_batch(batch)
    tr.weighted_loss(ls).backward()
    out = {}
    for dt in (torch.float32, torch.float64):
        p = {k: ((v.clone().to(dt) if v.is_floating_point() else v.clone()).requires_grad_(True) if v.is_floating_point() and 'running_' not in k else v.clone()) for k, v in sd.items()}
        bn = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in p.items() if 'running_' in k}
        try:
            ob = R.batch_from_arrays(lambda k: arrays[k])
            ols = R.run_batch(p, ctype, ob, training=True, bn_state=bn, p_drop=0.0, s_rounds=rounds, t_rounds=rounds)
            R.weighted_loss(ols, list(weights)).backward()
            out[dt] = ({k: float(ols[k].detach()) for k in ('recon_loss', 'prob_loss', 'func_loss')}, {k: p[k].grad for k in p if getattr(p[k], 'grad', None) is not None})
        except Exception as e:
            print('oracle in', dt, 'failed:', repr(e)[:200])
    print({k: float(ls[k]) for k in ('recon_loss', 'prob_loss', 'func_loss')})
    for dt, (l, g) in out.items():
        print(dt, l)
    g32 = out[torch.float32][1]
    g64 = out.get(torch.float64, (None, None))[1]
    for k, q in model.named_parameters():
        if q.grad is None or k not in g32: continue
        a = q.grad.detach().cpu().double().numpy(); b = g32[k].double().numpy()
        sc = max(1e-12, np.abs(b).max())
        line = '%-50s scale %.2e  hip-vs-f32 %.2e' % (k, sc, np.abs(a - b).max() / sc)
        if g64 is not None and k in g64:
            c = g64[k].numpy(); line += '  hip-vs-f64 %.2e  f32-vs-f64 %.2e' % (np.abs(a - c).max() / sc, np.abs(b - c).max() / sc)
        print(line)

gates = [('AND', [0, 1])] + [('NOT', [40 + i]) for i in range(300)]
w = (1.0, 0.0, 4.0) if len(sys.argv) > 1 and sys.argv[1] == 'func' else (1.0, 4.0, 0.0)
run('aig', [T.hand_graph('aig', 40, gates, seed=10)], w, 1)
