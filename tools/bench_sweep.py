#!/usr/bin/env python3
"""Level sweep in isolation on a config-2-shaped batch: HIP-event time of the forward and backward launcher calls
(nothing else on the GPU), same box A/B with MGV_LIB.

  python tools/bench_sweep.py [graphs=64] [iters=5]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multi-gate-vae_amd'))
import torch  # noqa: E402

import deepgate  # noqa: E402
from deepgate import ops, synthetic as syn  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    cfg = int(os.environ.get('CFG', '2'))
    dev = torch.device('cuda:0')
    arrays = syn.make_batch(cfg, batch=B)
    batch = deepgate.CircuitBatch.from_arrays(arrays, device=dev)
    gates = {2: [1, 2]}.get(cfg)
    if gates is None:
        ctype = {3: 'mig', 5: 'xmg'}[cfg]
        gates = [g for _, g in getattr(deepgate, 'dg_ae_model_' + ctype).Model.GATES]
    plan = deepgate.data.plan_of(batch, gates)
    N, H, T = plan.N, 64, len(gates)
    torch.manual_seed(0)
    hs = torch.randn(N, H, device=dev, requires_grad=True)
    attn_u = (torch.randn(T, 2 * H, device=dev) * 0.1).requires_grad_(True)
    Wvc = (torch.randn(T, 3 * H, 2 * H, device=dev) * 0.1).requires_grad_(True)
    bvc, bih, bhh = ((torch.randn(T, 3 * H, device=dev) * 0.1).requires_grad_(True) for _ in range(3))
    g = torch.randn(N, H, device=dev)

    def ev():
        return torch.cuda.Event(enable_timing=True)

    from deepgate import _hip
    tf = tb = 0.0
    kf, kb = [], []
    for it in range(iters + 1):
        _hip.profile(True)
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        hf = ops.FuncSweepFn.apply(plan, hs, attn_u, Wvc, bvc, bih, bhh)
        e1.record()
        hf.backward(g)
        e2.record()
        torch.cuda.synchronize()
        table = _hip.profile(False)
        if it > 0:
            tf += e0.elapsed_time(e1)
            tb += e1.elapsed_time(e2)
            kf += _hip.profile_times(table, 'mgv_func_sweep_fwd_x3')
            kb += _hip.profile_times(table, 'mgv_func_sweep_bwd_x3')
    print('N=%d levels=%d tiles=%d: sweep forward %.3f ms, backward %.3f ms; launcher calls alone: forward %.3f ms (min %.3f), backward %.3f ms (min %.3f)'
          % (N, plan.num_levels, plan.num_tiles, tf / iters, tb / iters, sum(kf) / len(kf), min(kf), sum(kb) / len(kb), min(kb)))


if __name__ == '__main__':
    main()
