#!/usr/bin/env python3
"""Diagnostic: where does a wave of the bf16x3 struct-stage kernels spend its cycles?
Runs the stamped build (csrc/libmgvae_diag.so, `make -C multi-gate-vae_amd/csrc diag`) of the forward
(and backward) kernel on a config-2-shaped batch and prints per-phase shares of the summed wave time.
Never quote this build's run time (its fences forbid overlaps the real kernel has)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multi-gate-vae_amd'))
import torch  # noqa: E402

import deepgate  # noqa: E402
from deepgate import ops, synthetic as syn  # noqa: E402
from deepgate.graph_plan import GraphPlan  # noqa: E402

FWD = ['row phase', 'barrier', 'prefetch+mfma', 'barrier', 'gru epilogue', 'idx commit', 'barrier', 'layernorm+store']
BWD = ['A row phase', 'barrier', 'B recompute mfma', 'B gate epilogue+commit', 'barrier', 'C ln stats (+barrier)', 'D ln/gru backward',
       '(unused)', 'F barrier', 'F output stage', 'E write dG tiles', 'E barrier', 'E dgrad mfma', 'E wgrad mfma', 'E pass-entry barrier']


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    dev = torch.device('cuda:0')
    lib = ctypes.CDLL(os.path.join(ROOT, 'multi-gate-vae_amd', 'csrc', 'libmgvae_diag.so'))
    arrays = syn.make_batch(2, batch=B)
    ei = torch.from_numpy(arrays['edge_index']).to(dev)
    N, H = arrays['num_nodes'], 64
    plan = GraphPlan(ei, N)
    torch.manual_seed(0)
    h = torch.randn(N, H, device=dev)
    xcls = torch.from_numpy(arrays['x'][:, 1].astype('uint8')).to(dev)
    xtab = torch.randn(6, 3 * H, device=dev) * 0.1
    Wc, Whh = torch.randn(3 * H, H, device=dev) * 0.1, torch.randn(3 * H, H, device=dev) * 0.1
    bc, bhh = torch.randn(3 * H, device=dev) * 0.1, torch.randn(3 * H, device=dev) * 0.1
    lw, lb = torch.ones(H, device=dev), torch.zeros(H, device=dev)
    wpack = ops.stage_wpack(Wc, Whh)
    out = torch.empty_like(h)
    stamps = torch.zeros(8 * 16, dtype=torch.int64, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.mgv_diag_set_stamps(P(stamps))
    for rev in (False, True):
        ptr, idx = plan.csr(rev)
        stamps.zero_()
        rc = lib.mgv_diag_struct_stage_fwd_x3_impl(H, ctypes.c_int64(N), P(h), P(ptr), P(idx), P(xcls), P(xtab), 6, P(wpack), P(bc), P(bhh),
                                                   P(lw), P(lb), ctypes.c_float(1e-5), P(out), 0, None, None, None, 0, None, st)
        torch.cuda.synchronize()
        assert rc == 0
        t = stamps.view(8, 16).double().cpu()
        tot = t.sum()
        print('forward, %s CSR: shares of summed wave cycles' % ('out' if rev else 'in'))
        for k, name in enumerate(FWD):
            print('   %-18s %5.1f%%   (per wave: %s)' % (name, 100 * t[:, k].sum() / tot, ' '.join('%4.1f' % (100 * v / tot * 8) for v in t[:, k])))
        # backward of the same half round
        gy = torch.randn(N, H, device=dev)
        gd, ga = torch.empty_like(h), torch.empty_like(h)
        acc = [torch.zeros(3 * H, H, device=dev), torch.zeros(3 * H, device=dev), torch.zeros(3 * H, H, device=dev),
               torch.zeros(3 * H, device=dev), torch.zeros(6, 3 * H, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)]
        stamps.zero_()
        rc = lib.mgv_diag_struct_stage_bwd_x3_impl(H, ctypes.c_int64(N), P(h), P(ptr), P(idx), P(xcls), P(xtab), 6, P(wpack), P(bc), P(bhh),
                                                   P(lw), P(lb), ctypes.c_float(1e-5), P(gy), P(out), P(gd), P(ga), *[P(t_) for t_ in acc], 0, None, None, None, 0, st)
        torch.cuda.synchronize()
        assert rc == 0
        t = stamps.view(8, 16).double().cpu()
        tot = t.sum()
        print('backward, %s CSR:' % ('out' if rev else 'in'))
        for k, name in enumerate(BWD):
            print('   %-26s %5.1f%%' % (name, 100 * t[:, k].sum() / tot))


if __name__ == '__main__':
    main()
