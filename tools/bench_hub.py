#!/usr/bin/env python3
"""Struct-stage kernels on a config-2-shaped batch with one HIGH-FAN-OUT node added (a reset/clock-like net): time per launch on
the out-CSR (where the hub's list is one row's neighbour list) against the same batch without the hub.

  python tools/bench_hub.py [fanout=100000] [graphs=16]      (MGV_HEAVY=0: without the heavy-row pre-pass)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multi-gate-vae_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from deepgate import _hip, ops, synthetic as syn  # noqa: E402
from deepgate._hip import ptr  # noqa: E402
from deepgate.graph_plan import GraphPlan  # noqa: E402


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    fan = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    dev = torch.device('cuda:0')
    arrays = syn.make_batch(2, batch=B)
    N, H, C = arrays['num_nodes'], 64, 6
    rng = np.random.Generator(np.random.PCG64(1))
    for label, extra in (('plain', 0), ('hub fan-out %d' % fan, fan)):
        ei = arrays['edge_index']
        if extra:
            dst = rng.choice(np.arange(1000, N), size=extra, replace=False)
            ei = np.concatenate([ei, np.stack([np.full(extra, 7, dtype=ei.dtype), dst.astype(ei.dtype)])], axis=1)
        plan = GraphPlan(torch.from_numpy(ei).to(dev), N)
        torch.manual_seed(0)
        h = torch.randn(N, H, device=dev)
        xcls = torch.from_numpy(arrays['x'][:, 1].astype('uint8')).to(dev)
        xtab = torch.randn(C, 3 * H, device=dev) * 0.1
        Wc, Whh = torch.randn(3 * H, H, device=dev) * 0.1, torch.randn(3 * H, H, device=dev) * 0.1
        bc, bhh = torch.randn(3 * H, device=dev) * 0.1, torch.randn(3 * H, device=dev) * 0.1
        lw, lb = torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev) * 0.1
        wpack = ops.stage_wpack(Wc, Whh)
        out = torch.empty_like(h)
        gy, ga = torch.randn(N, H, device=dev), torch.randn(N, H, device=dev)
        ws = ops._stage_ws(H, N, dev)
        p, i = plan.csr(True)
        hv = ops._heavy_args(plan.heavy(True) if os.environ.get('MGV_HEAVY', '1') != '0' else None, H, dev)
        acc = [torch.zeros(3 * H, H, device=dev), torch.zeros(3 * H, device=dev), torch.zeros(3 * H, H, device=dev),
               torch.zeros(3 * H, device=dev), torch.zeros(C, 3 * H, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)]
        gd, gg = torch.empty_like(h), torch.empty_like(h)
        stats = torch.empty(N, 2, device=dev)
        tf = timed(lambda: _hip.call('mgv_struct_stage_fwd_x3', H, N, ptr(h), ptr(p), ptr(i), ptr(xcls), ptr(xtab), C, ptr(wpack), ptr(bc), ptr(bhh),
                                     ptr(lw), ptr(lb), 1e-5, ptr(out), *hv, None, 0, ptr(stats)))
        tb = timed(lambda: _hip.call('mgv_struct_stage_bwd2_x3', H, N, ptr(h), ptr(p), ptr(i), ptr(xcls), ptr(xtab), C, ptr(wpack), ptr(bc), ptr(bhh),
                                     ptr(lw), ptr(lb), 1e-5, ptr(gy), ptr(ga), ptr(gd), ptr(gg), *[ptr(t) for t in acc], ptr(ws), ws.numel(), *hv, None, 0, ptr(stats)))
        print('%-24s N=%d E=%d: out-CSR forward %.3f ms, backward %.3f ms' % (label, N, ei.shape[1], tf, tb))


if __name__ == '__main__':
    main()
