#!/usr/bin/env python3
"""Diagnostic: per-phase shares of the summed wave cycles of k_struct_stage_bwd2_x3 (stamped build,
`make -C multi-gate-vae_amd/csrc diag`) on a config-2-shaped batch.  Read the SHARES, never this build's run time."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multi-gate-vae_amd'))
import torch  # noqa: E402

from deepgate import ops, synthetic as syn  # noqa: E402
from deepgate.graph_plan import GraphPlan  # noqa: E402

PH = ['P0 rows (+ barrier 0 behind the gather)', 'bar1', 'P1 recompute+exchange', 'bar2', 'P2 gru fwd + ln partials', 'bar3',
      'P3 ln/gru bwd + planes', 'bar4', 'P4a wgrad', 'P4b dgrad', '(unused)', '(unused)', '(unused)', 'P5 output stores']


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device('cuda:0')
    lib = ctypes.CDLL(os.path.join(ROOT, 'multi-gate-vae_amd', 'csrc', 'libmgvae_diag.so'))
    arrays = syn.make_batch(2, batch=B)
    ei = torch.from_numpy(arrays['edge_index']).to(dev)
    N, H, C = arrays['num_nodes'], 64, 6
    plan = GraphPlan(ei, N)
    torch.manual_seed(0)
    h = torch.randn(N, H, device=dev)
    xcls = torch.from_numpy(arrays['x'][:, 1].astype('uint8')).to(dev)
    xtab = torch.randn(C, 3 * H, device=dev) * 0.1
    Wc, Whh = torch.randn(3 * H, H, device=dev) * 0.1, torch.randn(3 * H, H, device=dev) * 0.1
    bc, bhh = torch.randn(3 * H, device=dev) * 0.1, torch.randn(3 * H, device=dev) * 0.1
    lw, lb = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    wpack = ops.stage_wpack(Wc, Whh)
    gy, ga_in = torch.randn(N, H, device=dev), torch.randn(N, H, device=dev)
    gd, ga = torch.empty_like(h), torch.empty_like(h)
    stamps = torch.zeros(8 * 16, dtype=torch.int64, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.mgv_diag_set_stamps2(P(stamps))
    n_ws = lib.mgv_diag_struct_stage_bwd2_ws_floats(H, ctypes.c_int64(N))
    ws = torch.empty(n_ws, device=dev)
    for rev in (False, True):
        ptr, idx = plan.csr(rev)
        acc = [torch.zeros(3 * H, H, device=dev), torch.zeros(3 * H, device=dev), torch.zeros(3 * H, H, device=dev),
               torch.zeros(3 * H, device=dev), torch.zeros(C, 3 * H, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)]
        stamps.zero_()
        rc = lib.mgv_diag_struct_stage_bwd2_x3_impl(H, ctypes.c_int64(N), P(h), P(ptr), P(idx), P(xcls), P(xtab), C, P(wpack), P(bc), P(bhh),
                                                    P(lw), P(lb), ctypes.c_float(1e-5), P(gy), P(ga_in), P(gd), P(ga), *[P(t_) for t_ in acc],
                                                    P(ws), ctypes.c_int64(n_ws), 0, None, None, None, 0, None, st)
        torch.cuda.synchronize()
        assert rc == 0, rc
        t = stamps.view(8, 16).double().cpu()
        tot = t.sum()
        ntile_per_wg = (N / 64) / 256
        print('bwd2, %s CSR: shares of summed wave cycles; cycles per tile per wave = %.0f' % ('out' if rev else 'in', float(tot) / 8 / 256 / ntile_per_wg))
        for k, name in enumerate(PH):
            print('   %-28s %5.1f%%   m=0 waves %5.1f%%  m=1 waves %5.1f%%' % (name, 100 * t[:, k].sum() / tot, 100 * t[:4, k].sum() / tot * 2, 100 * t[4:, k].sum() / tot * 2))


if __name__ == '__main__':
    main()
