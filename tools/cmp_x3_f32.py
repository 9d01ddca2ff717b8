"""Diagnostic: bf16x3 vs exact-f32 struct-stage kernels on one fixture graph (max abs differences)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multi-gate-vae_amd'))
from deepgate import ops
from deepgate.graph_plan import GraphPlan

z = np.load(os.path.join(ROOT, 'tests', 'golden', sys.argv[1] if len(sys.argv) > 1 else 'g2_xmg.npz'))
dev = torch.device('cuda:0')
ei = torch.tensor(z['in_edge_index'], device=dev)
N, H = z['in_x'].shape[0], 64
plan = GraphPlan(ei, N)
torch.manual_seed(1)
h = torch.randn(N, H, device=dev)
xcls = torch.tensor(z['in_x'][:, 1].astype('uint8'), device=dev)
xtab = torch.randn(6, 3 * H, device=dev) * 0.3
Wc, Whh = torch.randn(3 * H, H, device=dev) * 0.2, torch.randn(3 * H, H, device=dev) * 0.2
bc, bhh = torch.randn(3 * H, device=dev) * 0.1, torch.randn(3 * H, device=dev) * 0.1
lw, lb = torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev) * 0.1
gy, ga_in = torch.randn(N, H, device=dev), torch.randn(N, H, device=dev)
for rev in (False, True):
    p, i = plan.csr(rev)
    deg = (p[1:] - p[:-1])
    print('CSR', 'out' if rev else 'in', 'max degree', int(deg.max()))
    res = {}
    for prec in ('f32', 'x3'):
        ops.PRECISION = prec
        out = ops.struct_stage_fwd(h, p, i, xcls, xtab, Wc, bc, Whh, bhh, lw, lb)
        g = {'dxtab': torch.zeros_like(xtab), 'dWc': torch.zeros_like(Wc), 'dbc': torch.zeros_like(bc), 'dWhh': torch.zeros_like(Whh),
             'dbhh': torch.zeros_like(bhh), 'dln_w': torch.zeros_like(lw), 'dln_b': torch.zeros_like(lb)}
        gd, gg = ops.struct_stage_bwd(h, p, i, xcls, xtab, Wc, bc, Whh, bhh, lw, lb, gy, ga_in, g)
        torch.cuda.synchronize()
        res[prec] = dict(out=out, gd=gd, gg=gg, **g)
    for k in res['f32']:
        a, b = res['f32'][k], res['x3'][k]
        d = (a - b).abs()
        print('   %-6s max|f32| %.3e  max diff %.3e  rel %.2e  rows with diff>1e-3*max: %d' % (
            k, float(a.abs().max()), float(d.max()), float(d.max() / a.abs().max()),
            int((d.reshape(d.shape[0], -1).max(1).values > 1e-3 * a.abs().max()).sum())))
