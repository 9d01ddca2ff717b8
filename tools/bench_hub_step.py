#!/usr/bin/env python3
"""Whole train step on a config-2-shaped batch with one high-fan-out primary input (extra edges from node 7 to random gates):
step time and per-launcher device time against the same batch without it.

  python tools/bench_hub_step.py [fanout=10000] [graphs=16]
"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multi-gate-vae_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import deepgate  # noqa: E402
from deepgate import _hip, synthetic as syn  # noqa: E402


def main():
    fan = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    dev = torch.device('cuda:0')
    base = syn.make_batch(2, batch=B)
    N = base['num_nodes']
    rng = np.random.Generator(np.random.PCG64(1))
    for label, extra in (('plain', 0), ('hub fan-out %d' % fan, fan)):
        arrays = dict(base)
        if extra:
            gates = np.nonzero(base['forward_level'] > 0)[0]
            dst = rng.choice(gates, size=extra, replace=False)
            ei = base['edge_index']
            arrays['edge_index'] = np.concatenate([ei, np.stack([np.full(extra, 7, dtype=ei.dtype), dst.astype(ei.dtype)])], axis=1)
        batch = deepgate.CircuitBatch.from_arrays(arrays, device=dev)
        torch.manual_seed(0)
        enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=64, s_rounds=4, t_rounds=4, layernorm=True)
        model = deepgate.dg_ae_model_aig.Model(struct_encoder=enc, dim_hidden=64).to(dev).train()
        tr = deepgate.Trainer(types.SimpleNamespace(model='DG_AE'), model, training_id='hub', save_dir='/tmp/mgv_hub', lr=1e-4,
                              rc_prob_func_weight=[1.0, 4.0, 4.0], device='cuda:0', batch_size=B, distributed=False)
        for _ in range(2):
            tr.train_step(batch)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            tr.train_step(batch)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 3
        _hip.profile(True)
        tr.train_step(batch)
        table = _hip.profile(False)
        summ = _hip.profile_summary(table)
        top = sorted(summ.items(), key=lambda kv: -kv[1][1])[:7]
        print('%-22s N=%d E=%d: %.1f ms/step; %s' % (label, N, arrays['edge_index'].shape[1], ms,
                                                      ', '.join('%s %.1f' % (k.replace('mgv_', ''), v[1]) for k, v in top)))


if __name__ == '__main__':
    main()
