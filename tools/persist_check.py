#!/usr/bin/env python3
"""Persistent sweep kernels (csrc/sweep_persist_x3.hip) against the per-level kernels on BASELINE-shaped batches: results
(bit-identical forward; backward to rounding) and HIP-event times, same box, same run.

  python tools/persist_check.py [graphs=64] [iters=5] [fwd|both]      (CFG=2|3|5)
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
    mode = sys.argv[3] if len(sys.argv) > 3 else 'both'
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
    hs0 = torch.randn(N, H, device=dev)
    par0 = [torch.randn(T, 2 * H, device=dev) * 0.1, torch.randn(T, 3 * H, 2 * H, device=dev) * 0.1] + [torch.randn(T, 3 * H, device=dev) * 0.1 for _ in range(3)]
    g = torch.randn(N, H, device=dev)
    print('N=%d levels=%d tiles=%d roles=%s' % (N, plan.num_levels, plan.num_tiles, plan.persist_roles(ops._persist_ws(dev)[2])), flush=True)

    def ev():
        return torch.cuda.Event(enable_timing=True)

    res = {}
    for persist in (True, False):
        ops.PERSIST = persist
        tf = tb = 0.0
        for it in range(iters + 1):
            hs = hs0.clone().requires_grad_(True)
            par = [p.clone().requires_grad_(True) for p in par0]
            e0, e1, e2 = ev(), ev(), ev()
            e0.record()
            hf = ops.FuncSweepFn.apply(plan, hs, *par)
            e1.record()
            if mode == 'both':
                hf.backward(g)
            e2.record()
            torch.cuda.synchronize()
            if it > 0:
                tf += e0.elapsed_time(e1)
                tb += e1.elapsed_time(e2)
        ops.persist_check(dev)
        res[persist] = [hf.detach()] + ([hs.grad] + [p.grad for p in par] if mode == 'both' else [])
        print('%s: forward %.3f ms, backward %.3f ms' % ('persistent' if persist else 'per-level ', tf / iters, tb / iters), flush=True)
    names = ['hf', 'd hs', 'd attn_u', 'd Wvc', 'd bvc', 'd bih', 'd bhh']
    for nm, a, b in zip(names, res[True], res[False]):
        scale = float(b.abs().max())
        print('  %-9s max |diff| %.3e of scale %.3e  (%s)' % (nm, float((a - b).abs().max()), scale, 'bit-identical' if torch.equal(a, b) else 'differs'))


if __name__ == '__main__':
    main()
