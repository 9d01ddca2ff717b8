#!/usr/bin/env python3
"""Host-side cost of a train step (cProfile over N steps of a small config, where launches dominate): which Python layers the
~700 launches of a step go through.  python tools/host_profile.py [config=1] [steps=30]"""
import cProfile
import os
import pstats
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multi-gate-vae_amd'))
import torch  # noqa: E402

import deepgate  # noqa: E402
from deepgate import synthetic as syn  # noqa: E402


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    dev = torch.device('cuda:0')
    arrays = syn.make_batch(cfg)
    batch = deepgate.CircuitBatch.from_arrays(arrays, device=dev)
    torch.manual_seed(0)
    enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=64, s_rounds=4, t_rounds=4, layernorm=True)
    model = deepgate.dg_ae_model_aig.Model(struct_encoder=enc, dim_hidden=64).to(dev).train()
    tr = deepgate.Trainer(types.SimpleNamespace(model='DG_AE'), model, training_id='hp', save_dir='/tmp/mgv_hp', lr=1e-4,
                          rc_prob_func_weight=[1.0, 4.0, 4.0], device='cuda:0', batch_size=4, distributed=False)
    for _ in range(5):
        tr.train_step(batch)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        tr.enqueue_metrics(tr.train_step(batch))
    tr.flush_metrics()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(35)
    st.sort_stats('tottime').print_stats(25)


if __name__ == '__main__':
    main()
