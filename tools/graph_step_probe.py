#!/usr/bin/env python3
"""What would a whole-step HIP graph buy at the reference's own batch sizes?  Captures Trainer.train_step on a RESIDENT batch with
torch.cuda.graph and replays it.  A probe, not a product path: dropout / negative-sampling seeds and Adam's step count are by-value
kernel arguments, so every replay repeats the captured step's masks and bias correction.
  python tools/graph_step_probe.py [config=1] [batch=config's] [replays=100]"""
import contextlib
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multi-gate-vae_amd'))
import torch  # noqa: E402

import deepgate  # noqa: E402
from deepgate import synthetic as syn  # noqa: E402


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    B = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else None
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    dev = torch.device('cuda:0')
    arrays = syn.make_batch(cfg, batch=B) if B else syn.make_batch(cfg)
    batch = deepgate.CircuitBatch.from_arrays(arrays, device=dev)
    del batch.neg_edge_index
    torch.manual_seed(0)
    enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=64, s_rounds=4, t_rounds=4, layernorm=True)
    model = deepgate.dg_ae_model_aig.Model(struct_encoder=enc, dim_hidden=64)
    with contextlib.redirect_stdout(sys.stderr):
        tr = deepgate.Trainer(types.SimpleNamespace(model='DG_AE'), model, training_id='gp', save_dir='/tmp/mgv_gp', lr=1e-4,
                              rc_prob_func_weight=[1.0, 4.0, 4.0], device='cuda:0', batch_size=4, distributed=False)
    model.train()
    for _ in range(5):
        tr.train_step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        tr.train_step(batch)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / reps * 1e3
    print('N = %d: eager %.3f ms per step' % (batch.x.shape[0], eager))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            tr.train_step(batch)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ls = tr.train_step(batch)
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t0) / reps * 1e3
    print('N = %d: graph replay %.3f ms per step (%.2fx); losses of the last replay: %s' %
          (batch.x.shape[0], graphed, eager / graphed, [round(float(ls[k]), 5) for k in ('recon_loss', 'prob_loss', 'func_loss')]))


if __name__ == '__main__':
    main()
