#!/usr/bin/env python3
"""cProfile of the MAIN thread over the prefetched fresh-batch loop (config 2): where does `next(batch)` spend its time?
  python tools/fresh_host_profile.py [steps=60] [workers=8]"""
import contextlib
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
from deepgate.prefetch import BatchPrefetcher  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device('cuda:0')
    B, H, rounds = 64, 64, 4
    graphs = syn.make_graphs(2, batch=B)
    torch.manual_seed(0)
    enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=H, s_rounds=rounds, t_rounds=rounds, layernorm=True)
    model = deepgate.dg_ae_model_aig.Model(struct_encoder=enc, dim_hidden=H)
    with contextlib.redirect_stdout(sys.stderr):
        tr = deepgate.Trainer(types.SimpleNamespace(model='DG_AE'), model, training_id='fhp', save_dir='/tmp/mgv_fhp', lr=1e-4,
                              rc_prob_func_weight=[1.0, 4.0, 4.0], device=str(dev), batch_size=B, distributed=False)
    model.train()
    gate_ids = [g for _, g in model.GATES]

    def chunks(n):
        for s_ in range(n):
            yield graphs[s_ % B:] + graphs[:s_ % B]
    pf = BatchPrefetcher(chunks(steps + 10), dev, gate_ids=gate_ids, workers=workers, skip=('neg_edge_index',))
    it = iter(pf)
    for _ in range(10):
        tr.enqueue_metrics(tr.train_step(next(it)))
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        b = next(it)
        tr.enqueue_metrics(tr.train_step(b))
    tr.flush_metrics()
    torch.cuda.synchronize()
    pr.disable()
    pf.close()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(28)


if __name__ == '__main__':
    main()
