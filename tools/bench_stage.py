#!/usr/bin/env python3
"""Struct-stage kernels in isolation on a config-2-shaped batch: HIP-event time per launch of the forward kernel and of
both backward kernels (first decomposition / register-resident decomposition) on the in- and the out-CSR, plus the
largest deviation between the two backward kernels' outputs.  Same box, same run: the only fair A/B on this pool
(devices differ by ~10 % in clocks).

  python tools/bench_stage.py [graphs=64] [iters=5]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multi-gate-vae_amd'))
import torch  # noqa: E402

from deepgate import _hip, ops, synthetic as syn  # noqa: E402
from deepgate._hip import ptr  # noqa: E402
from deepgate.graph_plan import GraphPlan  # noqa: E402


def timed(fn, iters):
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    cfg = int(os.environ.get('CFG', '2'))
    dev = torch.device('cuda:0')
    arrays = syn.make_batch(cfg, batch=B)
    ei = torch.from_numpy(arrays['edge_index']).to(dev)
    N, H, C = arrays['num_nodes'], 64, 6
    plan = GraphPlan(ei, N)
    torch.manual_seed(0)
    h = torch.randn(N, H, device=dev)
    xcls = torch.from_numpy(arrays['x'][:, 1].astype('uint8')).to(dev)
    xtab = torch.randn(C, 3 * H, device=dev) * 0.1
    Wc, Whh = torch.randn(3 * H, H, device=dev) * 0.1, torch.randn(3 * H, H, device=dev) * 0.1
    bc, bhh = torch.randn(3 * H, device=dev) * 0.1, torch.randn(3 * H, device=dev) * 0.1
    lw, lb = torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev) * 0.1
    wpack = ops.stage_wpack(Wc, Whh)
    out = torch.empty_like(h)
    stats = torch.empty(N, 2, device=dev)      # LayerNorm statistics the forward keeps for the second backward kernel
    gy, ga_in = torch.randn(N, H, device=dev), torch.randn(N, H, device=dev)
    ws = ops._stage_ws(H, N, dev)

    def grads():
        return [torch.zeros(3 * H, H, device=dev), torch.zeros(3 * H, device=dev), torch.zeros(3 * H, H, device=dev),
                torch.zeros(3 * H, device=dev), torch.zeros(C, 3 * H, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)]

    print('N=%d E=%d (cfg %d, %d graphs)' % (N, plan.E, cfg, B))
    for rev in (False, True):
        p, i = plan.csr(rev)
        tag = 'out-CSR' if rev else 'in-CSR '

        def fwd():
            _hip.call('mgv_struct_stage_fwd_x3', H, N, ptr(h), ptr(p), ptr(i), ptr(xcls), ptr(xtab), C, ptr(wpack), ptr(bc), ptr(bhh),
                      ptr(lw), ptr(lb), 1e-5, ptr(out), 0, None, None, None, 0, ptr(stats))

        res = {}

        def bwd(which, acc, gd, ga):
            common = (H, N, ptr(h), ptr(p), ptr(i), ptr(xcls), ptr(xtab), C, ptr(wpack), ptr(bc), ptr(bhh), ptr(lw), ptr(lb), 1e-5,
                      ptr(gy), ptr(ga_in), ptr(gd), ptr(ga), *[ptr(t) for t in acc])
            if which == 1:
                _hip.call('mgv_struct_stage_bwd_x3', *common, 0, None, None, None, 0)
            else:
                _hip.call('mgv_struct_stage_bwd%d_x3' % which, *common, ptr(ws), ws.numel(), 0, None, None, None, 0,
                          ptr(stats) if os.environ.get('STAGE_NO_STATS') != '1' else None)

        t_f = timed(fwd, iters)
        line = '%s fwd %.3f ms' % (tag, t_f)
        only2 = os.environ.get('STAGE_ONLY2') == '1'      # ablation builds: time the second backward only, no comparison
        for which in ((2,) if only2 else (1, 2)):
            acc, gd, ga = grads(), torch.empty_like(h), torch.empty_like(h)
            bwd(which, acc, gd, ga)
            torch.cuda.synchronize()
            res[which] = [t.clone() for t in acc] + [gd.clone(), ga.clone()]
            t_b = timed(lambda: bwd(which, grads(), gd, ga), iters)
            line += ' | bwd%d %.3f ms' % (which, t_b)
        print(line)
        if only2:
            continue
        names = ['dWc', 'dbc', 'dWhh', 'dbhh', 'dxtab', 'dlnw', 'dlnb', 'g_direct', 'g_agg']
        worst = 0.0
        for other in (2,):
            worst = 0.0
            for n, a1, a2 in zip(names, res[1], res[other]):
                scale = float(a1.abs().max()) + 1e-30
                err = float((a1 - a2).abs().max()) / scale
                worst = max(worst, err)
                if err > 1e-4:
                    print('   %-9s deviates: %.3g of its scale (%.3g)' % (n, err, scale))
            print('   largest bwd1-vs-bwd%d deviation: %.3g of the tensor scale' % (other, worst))
            # run-to-run determinism
            acc2, gd2, ga2 = grads(), torch.empty_like(h), torch.empty_like(h)
            bwd(other, acc2, gd2, ga2)
            torch.cuda.synchronize()
            same = all(torch.equal(x, y) for x, y in zip(acc2 + [gd2, ga2], res[other]))
            print('   bwd%d bit-identical on a second run: %s' % (other, same))


if __name__ == '__main__':
    main()
