"""Host-side operators of the DG_AE hot path: thin checked wrappers over the C ABI
(include/mgvae_hip.h) and the torch.autograd.Functions that stitch the hand-written forward and
backward kernels into the reference's Python operator surface.  PyTorch supplies device memory,
streams and the autograd tape; all arithmetic on [N,*] data happens in libmgvae_hip.so.
"""
import torch

from . import _hip
from ._hip import check, ptr

F32 = torch.float32
I32 = torch.int32
U8 = torch.uint8
LN_EPS = 1e-5


def _zeros_like_params(*ts):
    return [torch.zeros_like(t) for t in ts]


# ------------------------------------------------------------------------------------------------
# structural encoder half round (digae_layer.py:267-275)
# ------------------------------------------------------------------------------------------------
def struct_stage_fwd(h_in, nbr_ptr, nbr_idx, xcls, xtab, Wc, bc, Whh, bhh, ln_w, ln_b, out=None):
    N, H = h_in.shape
    check(h_in, F32, 'h_in'); check(nbr_ptr, I32, 'nbr_ptr'); check(nbr_idx, I32, 'nbr_idx'); check(xcls, U8, 'xcls')
    for n, t in (('xtab', xtab), ('Wc', Wc), ('bc', bc), ('Whh', Whh), ('bhh', bhh)):
        check(t, F32, n)
    check(ln_w, F32, 'ln_w'); check(ln_b, F32, 'ln_b')
    assert nbr_ptr.numel() == N + 1 and xcls.numel() == N
    assert Wc.shape == (3 * H, H) and Whh.shape == (3 * H, H) and xtab.shape[1] == 3 * H
    h_out = torch.empty_like(h_in) if out is None else out
    _hip.call('mgv_struct_stage_fwd', H, N, ptr(h_in), ptr(nbr_ptr), ptr(nbr_idx), ptr(xcls), ptr(xtab),
              xtab.shape[0], ptr(Wc), ptr(bc), ptr(Whh), ptr(bhh), ptr(ln_w), ptr(ln_b), LN_EPS, ptr(h_out))
    return h_out


def struct_stage_bwd(h_in, nbr_ptr, nbr_idx, xcls, xtab, Wc, bc, Whh, bhh, ln_w, ln_b, gy_direct, gy_agg,
                     grads, need_input_grad=True):
    """`grads` = dict of fp32 accumulators (dWc, dbc, dWhh, dbhh, dxtab, dln_w, dln_b), added to."""
    N, H = h_in.shape
    check(gy_direct, F32, 'gy_direct'); check(gy_agg, F32, 'gy_agg')
    WcT = Wc.t().contiguous()
    WhhT = Whh.t().contiguous()
    g_direct = torch.empty_like(h_in) if need_input_grad else None
    g_agg = torch.empty_like(h_in) if need_input_grad else None
    _hip.call('mgv_struct_stage_bwd', H, N, ptr(h_in), ptr(nbr_ptr), ptr(nbr_idx), ptr(xcls), ptr(xtab),
              xtab.shape[0], ptr(Wc), ptr(WcT), ptr(bc), ptr(Whh), ptr(WhhT), ptr(bhh), ptr(ln_w), ptr(ln_b),
              LN_EPS, ptr(gy_direct), ptr(gy_agg), ptr(g_direct), ptr(g_agg), ptr(grads['dWc']), ptr(grads['dbc']),
              ptr(grads['dWhh']), ptr(grads['dbhh']), ptr(grads['dxtab']), ptr(grads.get('dln_w')),
              ptr(grads.get('dln_b')))
    return g_direct, g_agg


class StructEncoderFn(torch.autograd.Function):
    """All 2R half rounds of one MultiGCNEncoder (digae_layer.py:257-277) as one autograd node.

    Forward keeps the input state of every half round (2R x [N,H]); backward walks them in reverse
    and hands each stage's aggregate gradient to the previous stage's gather (consecutive half
    rounds use opposite CSRs), so no scatter pass exists.
    Inputs: the composed per-direction weights (forward half: *_f, reversed half: *_r) and the shared
    LayerNorm affine (None, None = no LayerNorm)."""

    @staticmethod
    def forward(ctx, plan, xcls, rounds, xtab_f, Wc_f, bc_f, Whh_f, bhh_f, xtab_r, Wc_r, bc_r, Whh_r, bhh_r, ln_w, ln_b):
        N = plan.N
        H = Whh_f.shape[1]
        dev = Whh_f.device
        par = [t.detach().contiguous() for t in (xtab_f, Wc_f, bc_f, Whh_f, bhh_f, xtab_r, Wc_r, bc_r, Whh_r, bhh_r)]
        lw = ln_w.detach().contiguous() if ln_w is not None else None
        lb = ln_b.detach().contiguous() if ln_b is not None else None
        h = torch.ones(N, H, dtype=F32, device=dev)          # node_state = ones (digae_layer.py:260)
        states = []
        for _ in range(rounds):
            for rev in (False, True):
                p, i = plan.csr(rev)
                w = par[5:] if rev else par[:5]
                states.append(h)
                h = struct_stage_fwd(h, p, i, xcls, w[0], w[1], w[2], w[3], w[4], lw, lb)
        ctx.plan, ctx.xcls, ctx.rounds = plan, xcls, rounds
        ctx.par, ctx.lw, ctx.lb, ctx.states = par, lw, lb, states
        return h

    @staticmethod
    def backward(ctx, gy):
        plan, xcls, par, lw, lb = ctx.plan, ctx.xcls, ctx.par, ctx.lw, ctx.lb
        gy = gy.contiguous()
        acc = {}
        for tag, w in (('f', par[:5]), ('r', par[5:])):
            acc[tag] = {'dxtab': torch.zeros_like(w[0]), 'dWc': torch.zeros_like(w[1]), 'dbc': torch.zeros_like(w[2]),
                        'dWhh': torch.zeros_like(w[3]), 'dbhh': torch.zeros_like(w[4])}
        dlw = torch.zeros_like(lw) if lw is not None else None
        dlb = torch.zeros_like(lb) if lb is not None else None
        g_direct, g_agg = gy, None
        k = len(ctx.states) - 1
        for _ in range(ctx.rounds):
            for rev in (True, False):
                p, i = plan.csr(rev)
                w = par[5:] if rev else par[:5]
                g = dict(acc['r' if rev else 'f'])
                g['dln_w'], g['dln_b'] = dlw, dlb
                g_direct, g_agg = struct_stage_bwd(ctx.states[k], p, i, xcls, w[0], w[1], w[2], w[3], w[4], lw, lb,
                                                   g_direct, g_agg, g, need_input_grad=(k > 0))
                k -= 1
        ctx.states = None
        f, r = acc['f'], acc['r']
        return (None, None, None, f['dxtab'], f['dWc'], f['dbc'], f['dWhh'], f['dbhh'],
                r['dxtab'], r['dWc'], r['dbc'], r['dWhh'], r['dbhh'], dlw, dlb)
