#!/usr/bin/env python3
"""Where does the fresh-batch loop lose its ~3 ms per step against the resident-batch loop?  Host time waiting for the next batch,
host time to enqueue a step, and the wall time per step, for the resident batch and for prefetched fresh batches.
  python tools/fresh_probe.py [steps=12] [workers=4]"""
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
from deepgate.data import plan_of  # noqa: E402
from deepgate.prefetch import BatchPrefetcher  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device('cuda:0')
    B, H, rounds = 64, 64, 4
    graphs = syn.make_graphs(2, batch=B)
    batch = deepgate.CircuitBatch.from_arrays(syn.collate(graphs), device=dev)
    del batch.neg_edge_index
    torch.manual_seed(0)
    enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=H, s_rounds=rounds, t_rounds=rounds, layernorm=True)
    model = deepgate.dg_ae_model_aig.Model(struct_encoder=enc, dim_hidden=H)
    with contextlib.redirect_stdout(sys.stderr):
        tr = deepgate.Trainer(types.SimpleNamespace(model='DG_AE'), model, training_id='probe', save_dir='/tmp/mgv_probe', lr=1e-4,
                              rc_prob_func_weight=[1.0, 4.0, 4.0], device=str(dev), batch_size=B, distributed=False)
    model.train()
    gate_ids = [g for _, g in model.GATES]
    plan_of(batch, gate_ids)

    def loop(next_batch, n):
        t_next = t_step = 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            a = time.perf_counter()
            b = next_batch()
            c = time.perf_counter()
            tr.enqueue_metrics(tr.train_step(b))
            d = time.perf_counter()
            t_next += c - a
            t_step += d - c
        tr.flush_metrics()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return el / n * 1e3, t_next / n * 1e3, t_step / n * 1e3

    if os.environ.get('HIGH', '0') == '1':          # the step on a high-priority stream (the device offers normal and high only)
        hp = torch.cuda.Stream(priority=-1)
        tr._side = torch.cuda.Stream(priority=-1)
        torch.cuda.set_stream(hp)
    loop(lambda: batch, 3)
    print('resident: %.2f ms/step wall, %.2f ms waiting for the batch, %.2f ms host enqueue' % loop(lambda: batch, steps))

    def chunks(n):
        for s_ in range(n):
            yield graphs[s_ % B:] + graphs[:s_ % B]
    for mode in ('plan on the worker', 'copy only (plan in the step)'):
        pf = BatchPrefetcher(chunks(steps + 3), dev, gate_ids=gate_ids if mode.startswith('plan') else None, workers=workers, skip=('neg_edge_index',))
        it = iter(pf)
        loop(lambda: next(it), 3)
        print('fresh, %s: %.2f ms/step wall, %.2f ms waiting for the batch, %.2f ms host enqueue' % ((mode,) + loop(lambda: next(it), steps)))
        pf.close()
    # fresh batches whose host side is free: the same device batch object re-planned every step on a side stream is not possible
    # without the prefetcher; instead time the worker-side work alone
    pf = BatchPrefetcher(chunks(steps), dev, gate_ids=gate_ids, workers=workers, skip=('neg_edge_index',))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = sum(1 for _ in pf)
    torch.cuda.synchronize()
    print('prefetcher alone (collate + H2D + plan + caches, %d workers): %.2f ms per batch' % (workers, (time.perf_counter() - t0) / n * 1e3))
    pf.close()


if __name__ == '__main__':
    main()
