"""Diagnostic: run one train step with torch.empty() poisoned (NaN / INT_MAX) to expose reads of memory no kernel wrote."""
import os, sys, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multi-gate-vae_amd'))
import torch
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
import deepgate
from deepgate import synthetic as syn
dev = torch.device('cuda:0')
torch.manual_seed(0)
H, rounds = 64, 2
ctype = sys.argv[1] if len(sys.argv) > 1 else 'aig'
enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=H, s_rounds=rounds, t_rounds=rounds, layernorm=True)
model = getattr(deepgate, 'dg_ae_model_' + ctype).Model(struct_encoder=enc, dim_hidden=H)
graphs = [syn.make_graph(ctype, 512, 16, 77 + i, n_inputs=32) for i in range(2)]
batch = deepgate.CircuitBatch.from_arrays(syn.collate(graphs), device=dev)
tr = deepgate.Trainer(types.SimpleNamespace(model='DG_AE'), model, training_id='dbg', save_dir='/tmp/mgv_dbg', lr=1e-4,
                      rc_prob_func_weight=[1.0, 4.0, 4.0], device='cuda:0', batch_size=2, distributed=False)
model.train(); tr.optimizer.zero_grad()
ls = tr.run_batch(batch)
print({k: float(ls[k]) for k in ('recon_loss', 'prob_loss', 'func_loss')})
tr.weighted_loss(ls).backward()
bad = [k for k, v in model.named_parameters() if v.grad is not None and not bool(torch.isfinite(v.grad).all())]
print('parameters with non-finite gradients:', bad if bad else 'none')
