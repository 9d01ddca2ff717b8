"""Host-side operators of the DG_AE hot path: thin checked wrappers over the C ABI
(include/mgvae_hip.h) and the torch.autograd.Functions that stitch the hand-written forward and
backward kernels into the reference's Python operator surface.  PyTorch supplies device memory,
streams and the autograd tape; all arithmetic on [N,*] data happens in libmgvae_hip.so.
"""
import torch

from . import _hip
from ._hip import HipLibraryError, check, ptr

import os

F32 = torch.float32
I32 = torch.int32
U8 = torch.uint8
LN_EPS = 1e-5

# Arithmetic of the dense products: 'x3' = bf16x3 split-precision MFMA (hi/lo bf16 planes, three
# products, fp32 accumulation; ~1e-5 relative per product, 3/16 of the fp32-MFMA cost), 'f32' = exact
# fp32 MFMA.  Hidden widths the x3 kernels do not cover (H=16) always run in fp32.
PRECISION = os.environ.get('MGV_PRECISION', 'x3')
# first half round of an encoder from one kernel row per (degree, class) pair ('table') or over all nodes ('full')
FIRST_STAGE_TABLE = True          # first half round per (degree, class) pair / quotient stages (tests switch it off to compare with the per-node launch)


def use_x3(H):
    return PRECISION == 'x3' and H in (32, 64)


def split_bf16(w):
    hi = w.to(torch.bfloat16)
    lo = (w - hi.to(torch.float32)).to(torch.bfloat16)
    return hi, lo


def frag_order(w):
    """[R, K] (k contiguous) -> MFMA 16x16x32 fragment order: blocks (row tile, k-step) of 512 elements in
    which lane l = 16*q + r holds w[16*rt + r, 32*ks + 8*q : +8] at offset 8*l (one coalesced 1 KiB load)."""
    R, K = w.shape
    return w.reshape(R // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(-1)


def _pack_into(out, off, W, transpose):
    """bf16 hi/lo fragment-order planes of W (or W^T) at out[off : off + 2 * W.numel()] (one launch)."""
    R, K = (W.shape[1], W.shape[0]) if transpose else W.shape
    n = W.numel()
    if W.dtype != F32 or not W.is_cuda or W.stride(1) != 1:
        raise HipLibraryError('weight must be an fp32 GPU matrix with contiguous rows')
    base = out.data_ptr() + 2 * off
    _hip.call('mgv_wpack_bf16x3', ptr(W), R, K, W.stride(0), int(transpose), _hip.ctypes.c_void_p(base),
              _hip.ctypes.c_void_p(base + 2 * n))
    return off + 2 * n


def stage_wpack(Wc, Whh):
    """bf16 weight pack of mgv_struct_stage_*_x3: [Wc_hi, Wc_lo, Whh_hi, Whh_lo, WcT_hi, WcT_lo, WhhT_hi, WhhT_lo],
    each block in fragment order."""
    out = torch.empty(4 * (Wc.numel() + Whh.numel()), dtype=torch.bfloat16, device=Wc.device)
    off = 0
    for w, tr in ((Wc, False), (Whh, False), (Wc, True), (Whh, True)):
        off = _pack_into(out, off, w, tr)
    return out


def sweep_wpack(Wvc):
    """bf16 weight pack of mgv_func_sweep_*_x3 from Wvc [T, 3H, 2H]: per slot [Wvc_hi, Wvc_lo, WvcT_hi, WvcT_lo],
    each block in fragment order."""
    out = torch.empty(4 * Wvc.numel(), dtype=torch.bfloat16, device=Wvc.device)
    off = 0
    for g in range(Wvc.shape[0]):
        off = _pack_into(out, off, Wvc[g], False)
        off = _pack_into(out, off, Wvc[g], True)
    return out


def linear_wpack(W):
    out = torch.empty(2 * W.numel(), dtype=torch.bfloat16, device=W.device)
    _pack_into(out, 0, W, False)
    return out


def _zeros_like_params(*ts):
    return [torch.zeros_like(t) for t in ts]


# ------------------------------------------------------------------------------------------------
# structural encoder half round (digae_layer.py:267-275)
# ------------------------------------------------------------------------------------------------
def _heavy_args(heavy, H, device):
    """(heavy_n, heavy_nodes, heavy_ws) of a stage launch; `heavy` = GraphPlan.heavy(reverse) or None."""
    if heavy is None or heavy[0] == 0:
        return 0, None, None
    n, nodes = heavy
    hw = _WS.get(('heavy', str(device)))
    if hw is None or hw.numel() < 2 * n * H:
        hw = _WS[('heavy', str(device))] = torch.empty(2 * n * H, dtype=F32, device=device)
    return n, ptr(nodes), ptr(hw)


def struct_stage_fwd(h_in, nbr_ptr, nbr_idx, xcls, xtab, Wc, bc, Whh, bhh, ln_w, ln_b, out=None, wpack=None, heavy=None, table_own=None, n_rows=None,
                     stats_out=None, tagged=True):
    """`table_own` (int32 [N]): table mode — h_in is the (degree, class) table, nbr_idx entries are tagged (GraphPlan.tagged_idx);
    with `tagged=False` the entries are plain rows of h_in and only the own rows go through table_own (quotient stages: h_in is the
    previous stage's colour table).
    `n_rows`: only the first n_rows rows of h_in are stage rows, the rest are rows their neighbour lists point at (quotient stages).
    `stats_out` [N, 2] (bf16x3 kernels): receives {mean, rstd} of every row's pre-LayerNorm state, for struct_stage_bwd(stats=...)."""
    N, H = h_in.shape
    if n_rows is not None:
        N = int(n_rows)
    if table_own is not None:
        N = table_own.numel()
    check(h_in, F32, 'h_in'); check(nbr_ptr, I32, 'nbr_ptr'); check(nbr_idx, I32, 'nbr_idx'); check(xcls, U8, 'xcls')
    for n, t in (('xtab', xtab), ('Wc', Wc), ('bc', bc), ('Whh', Whh), ('bhh', bhh)):
        check(t, F32, n)
    check(ln_w, F32, 'ln_w'); check(ln_b, F32, 'ln_b')
    assert nbr_ptr.numel() == N + 1 and xcls.numel() == N
    assert Wc.shape == (3 * H, H) and Whh.shape == (3 * H, H) and xtab.shape[1] == 3 * H
    h_out = (torch.empty(N, H, dtype=F32, device=h_in.device) if out is None else out)
    if use_x3(H):
        wpack = stage_wpack(Wc, Whh) if wpack is None else wpack
        _hip.call('mgv_struct_stage_fwd_x3', H, N, ptr(h_in), ptr(nbr_ptr), ptr(nbr_idx), ptr(xcls), ptr(xtab),
                  xtab.shape[0], ptr(wpack), ptr(bc), ptr(bhh), ptr(ln_w), ptr(ln_b), LN_EPS, ptr(h_out), *_heavy_args(heavy, H, h_in.device), ptr(table_own),
                  int(bool(tagged)), ptr(stats_out))
        return h_out
    assert table_own is None, 'table mode needs the bf16x3 kernels'
    _hip.call('mgv_struct_stage_fwd', H, N, ptr(h_in), ptr(nbr_ptr), ptr(nbr_idx), ptr(xcls), ptr(xtab),
              xtab.shape[0], ptr(Wc), ptr(bc), ptr(Whh), ptr(bhh), ptr(ln_w), ptr(ln_b), LN_EPS, ptr(h_out))
    return h_out


# The bf16x3 half-round backward at H = 64 is struct_stage_bwd2_x3.hip (register-resident recompute weights, transposed products,
# slab-reduced deterministic parameter gradients); the first kernel (struct_stage_x3.hip) serves H = 32 only (tools/bench_stage.py
# still times both through the C ABI).
TABLE_MODE = True                 # half round 2 of an encoder reads the (degree, class) table directly
QUOTIENT = os.environ.get('MGV_QUOTIENT', '1') != '0'         # early half rounds on one row per colour (GraphPlan.quotient)
_WS = {}





def workspace(nfloats, device, dtype=F32):
    """Scratch for the deterministic cross-workgroup sums (per-workgroup partial rows, csrc/mgv_slab.h): one growing buffer
    per (device, stream, dtype); launches on a stream are ordered, so consecutive users may share it."""
    key = (str(device), _hip.stream().value if torch.device(device).type == 'cuda' else 0, dtype)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nfloats:
        buf = _WS[key] = torch.empty(max(int(nfloats), 1), dtype=dtype, device=device)
    return buf


def sum_ws(device):
    """Double workspace of the small-sum launchers (mgv_sum_workspace_doubles), per (device, stream)."""
    return workspace(_hip.call_value('mgv_sum_workspace_doubles'), device, torch.float64)


def _sw(device):
    ws = sum_ws(device)
    return ptr(ws), ws.numel()


def _stage_ws(H, N, device):
    """Scratch slab of mgv_struct_stage_bwd2_x3 (one per device, grown on demand; contents are never read across calls)."""
    return workspace(_hip.call_value('mgv_struct_stage_bwd2_ws_floats', H, N), device)


def struct_stage_bwd(h_in, nbr_ptr, nbr_idx, xcls, xtab, Wc, bc, Whh, bhh, ln_w, ln_b, gy_direct, gy_agg,
                     grads, need_input_grad=True, wpack=None, heavy=None, table_own=None, n_rows=None, stats=None, tagged=True):
    """`grads` = dict of fp32 accumulators (dWc, dbc, dWhh, dbhh, dxtab, dln_w, dln_b), added to."""
    N, H = h_in.shape
    if n_rows is not None:
        N = int(n_rows)
    if table_own is not None:
        N = table_own.numel()
    check(gy_direct, F32, 'gy_direct'); check(gy_agg, F32, 'gy_agg')
    if use_x3(H) and H == 64:
        g_direct = torch.empty(N, H, dtype=F32, device=h_in.device) if need_input_grad else None
        g_agg = torch.empty(N, H, dtype=F32, device=h_in.device) if need_input_grad else None
        wpack = stage_wpack(Wc, Whh) if wpack is None else wpack
        ws = _stage_ws(H, N, h_in.device)
        _hip.call('mgv_struct_stage_bwd2_x3', H, N, ptr(h_in), ptr(nbr_ptr), ptr(nbr_idx), ptr(xcls), ptr(xtab),
                  xtab.shape[0], ptr(wpack), ptr(bc), ptr(bhh), ptr(ln_w), ptr(ln_b), LN_EPS, ptr(gy_direct),
                  ptr(gy_agg), ptr(g_direct), ptr(g_agg), ptr(grads['dWc']), ptr(grads['dbc']), ptr(grads['dWhh']),
                  ptr(grads['dbhh']), ptr(grads['dxtab']), ptr(grads.get('dln_w')), ptr(grads.get('dln_b')),
                  ptr(ws), ws.numel(), *_heavy_args(heavy, H, h_in.device), ptr(table_own), int(bool(tagged)), ptr(stats))
        return g_direct, g_agg
    assert table_own is None or not tagged, 'table mode needs the H = 64 bf16x3 backward'
    if use_x3(H):
        g_direct = torch.empty(N, H, dtype=F32, device=h_in.device) if need_input_grad else None
        g_agg = torch.empty(N, H, dtype=F32, device=h_in.device) if need_input_grad else None
        wpack = stage_wpack(Wc, Whh) if wpack is None else wpack
        _hip.call('mgv_struct_stage_bwd_x3', H, N, ptr(h_in), ptr(nbr_ptr), ptr(nbr_idx), ptr(xcls), ptr(xtab),
                  xtab.shape[0], ptr(wpack), ptr(bc), ptr(bhh), ptr(ln_w), ptr(ln_b), LN_EPS, ptr(gy_direct),
                  ptr(gy_agg), ptr(g_direct), ptr(g_agg), ptr(grads['dWc']), ptr(grads['dbc']), ptr(grads['dWhh']),
                  ptr(grads['dbhh']), ptr(grads['dxtab']), ptr(grads.get('dln_w')), ptr(grads.get('dln_b')), *_heavy_args(heavy, H, h_in.device), ptr(table_own), 0)
        return g_direct, g_agg
    assert table_own is None, 'own rows through an index need the bf16x3 kernels'
    WcT = Wc.t().contiguous()
    WhhT = Whh.t().contiguous()
    g_direct = torch.empty(N, H, dtype=F32, device=h_in.device) if need_input_grad else None
    g_agg = torch.empty(N, H, dtype=F32, device=h_in.device) if need_input_grad else None
    _hip.call('mgv_struct_stage_bwd', H, N, ptr(h_in), ptr(nbr_ptr), ptr(nbr_idx), ptr(xcls), ptr(xtab),
              xtab.shape[0], ptr(Wc), ptr(WcT), ptr(bc), ptr(Whh), ptr(WhhT), ptr(bhh), ptr(ln_w), ptr(ln_b),
              LN_EPS, ptr(gy_direct), ptr(gy_agg), ptr(g_direct), ptr(g_agg), ptr(grads['dWc']), ptr(grads['dbc']),
              ptr(grads['dWhh']), ptr(grads['dbhh']), ptr(grads['dxtab']), ptr(grads.get('dln_w')),
              ptr(grads.get('dln_b')))
    return g_direct, g_agg


def _seg_sums(H, tables, items, direct, agg=None, nbr_ptr=None, nbr_idx=None):
    """Per-group row sums through the segment tables of GraphPlan.class_sum_levels: level 1 reads rows `items` of `direct` (+ the
    neighbour pull of `agg`), every further level the partial rows the one before left behind the C final rows of the same buffer."""
    buf = torch.empty(tables['rows'], H, dtype=F32, device=direct.device)
    for li, (n_seg, seg_ptr, out_row, src_row) in enumerate(tables['levels']):
        if li == 0:
            _hip.call('mgv_seg_sum', H, n_seg, ptr(seg_ptr), ptr(items), ptr(direct), ptr(agg), ptr(nbr_ptr), ptr(nbr_idx), ptr(out_row), ptr(buf))
        else:
            _hip.call('mgv_seg_sum', H, n_seg, ptr(seg_ptr), None, ptr(buf[src_row:]), None, None, None, ptr(out_row), ptr(buf))
    return buf[:tables['C']]


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
        packs = (stage_wpack(par[1], par[3]), stage_wpack(par[6], par[8])) if use_x3(H) else (None, None)
        # node_state = ones (digae_layer.py:260): the first half round sees identical rows, one kernel row per
        # (degree, feature class) pair does for all nodes of the pair
        # quotient stages: while the rows of a half round are few distinct ones, it runs on one representative row per colour
        quot = plan.quotient(xcls, 2 * rounds) if (QUOTIENT and FIRST_STAGE_TABLE and rounds > 0 and N > 0 and Whh_f.is_cuda) else []
        if quot:
            return StructEncoderFn._forward_quotient(ctx, plan, xcls, rounds, par, lw, lb, packs, quot)
        first = plan.first_stage_classes(xcls) if (FIRST_STAGE_TABLE and rounds > 0 and N > 0) else None
        h = None if first is not None else torch.ones(N, H, dtype=F32, device=dev)
        # table mode for the half round after the table one: bf16x3 H = 64 kernels, node ids and table rows fit a tagged 32-bit entry
        table_mode = first is not None and use_x3(H) and H == 64 and N < (1 << 24) and first[1] <= 256 and TABLE_MODE
        states, stats = [], []
        keep_stats = lw is not None and use_x3(H) and H == 64      # LayerNorm statistics kept for the bwd2 kernel
        for _ in range(rounds):
            for rev in (False, True):
                p, i = plan.csr(rev)
                w = par[5:] if rev else par[:5]
                states.append(h)
                stats.append(torch.empty((first[1] if h is None else N), 2, dtype=F32, device=dev) if keep_stats else None)
                if h is None:
                    cid, C, tp, ti, tx = first
                    table = struct_stage_fwd(torch.ones(C, H, dtype=F32, device=dev), tp, ti, tx, w[0], w[1], w[2], w[3], w[4],
                                             lw, lb, wpack=packs[0], stats_out=stats[-1])
                    if table_mode:
                        h = ('table', table)     # never expanded to N rows: the next half round reads the table through tagged entries
                    else:
                        h = torch.empty(N, H, dtype=F32, device=dev)
                        _hip.call('mgv_class_expand', H, N, ptr(table), ptr(cid), ptr(h))
                elif isinstance(h, tuple):
                    h = struct_stage_fwd(h[1], p, plan.tagged_idx(rev, first[0]), xcls, w[0], w[1], w[2], w[3], w[4], lw, lb, wpack=packs[int(rev)],
                                         heavy=plan.heavy(rev), table_own=first[0], stats_out=stats[-1])
                else:
                    h = struct_stage_fwd(h, p, i, xcls, w[0], w[1], w[2], w[3], w[4], lw, lb, wpack=packs[int(rev)], heavy=plan.heavy(rev),
                                         stats_out=stats[-1])
        ctx.plan, ctx.xcls, ctx.rounds, ctx.packs, ctx.first = plan, xcls, rounds, packs, first
        ctx.par, ctx.lw, ctx.lb, ctx.states, ctx.stats = par, lw, lb, states, stats
        ctx.quot = None
        return h

    @staticmethod
    def _forward_quotient(ctx, plan, xcls, rounds, par, lw, lb, packs, quot):
        """Half rounds 1..len(quot) on one row per colour (inputs: the previous stage's table, stacked behind the representatives'
        own rows), the table of the last one expanded to N rows, the remaining half rounds as usual."""
        N, H, dev = plan.N, par[3].shape[1], par[3].device
        states, stats = [], []
        keep_stats = lw is not None and use_x3(H) and H == 64
        table = torch.ones(1, H, dtype=F32, device=dev)
        h = None
        for k in range(2 * rounds):
            rev = k % 2 == 1
            w = par[5:] if rev else par[:5]
            if k < len(quot):
                st = quot[k]
                stats.append(torch.empty(st['C'], 2, dtype=F32, device=dev) if keep_stats else None)
                if use_x3(H):
                    # the bf16x3 kernels read the previous table in place: own rows through st['own32'], lists name its rows
                    states.append(table)
                    table = struct_stage_fwd(table, st['ptr'], st['ent_idx'], st['xcls'], w[0], w[1], w[2], w[3], w[4], lw, lb, wpack=packs[int(rev)],
                                             heavy=st['heavy'], table_own=st['own32'], tagged=False, stats_out=stats[-1])
                else:
                    h_cat = torch.cat([table.index_select(0, st['own']), table])
                    states.append(h_cat)
                    table = struct_stage_fwd(h_cat, st['ptr'], st['idx'], st['xcls'], w[0], w[1], w[2], w[3], w[4], lw, lb, wpack=packs[int(rev)],
                                             heavy=st['heavy'], n_rows=st['C'], stats_out=stats[-1])
                if k + 1 == len(quot) or k + 1 == 2 * rounds:
                    h = torch.empty(N, H, dtype=F32, device=dev)
                    _hip.call('mgv_class_expand', H, N, ptr(table), ptr(st['cid']), ptr(h))
            else:
                p, i = plan.csr(rev)
                states.append(h)
                stats.append(torch.empty(N, 2, dtype=F32, device=dev) if keep_stats else None)
                h = struct_stage_fwd(h, p, i, xcls, w[0], w[1], w[2], w[3], w[4], lw, lb, wpack=packs[int(rev)], heavy=plan.heavy(rev),
                                     stats_out=stats[-1])
        ctx.plan, ctx.xcls, ctx.rounds, ctx.packs, ctx.first = plan, xcls, rounds, packs, None
        ctx.par, ctx.lw, ctx.lb, ctx.states, ctx.stats = par, lw, lb, states, stats
        ctx.quot = quot
        return h

    @staticmethod
    def _backward_quotient(ctx, gy):
        plan, xcls, par, lw, lb, quot = ctx.plan, ctx.xcls, ctx.par, ctx.lw, ctx.lb, ctx.quot
        H, dev = gy.shape[1], gy.device
        acc = {}
        for tag, w in (('f', par[:5]), ('r', par[5:])):
            acc[tag] = {'dxtab': torch.zeros_like(w[0]), 'dWc': torch.zeros_like(w[1]), 'dbc': torch.zeros_like(w[2]),
                        'dWhh': torch.zeros_like(w[3]), 'dbhh': torch.zeros_like(w[4])}
        dlw = torch.zeros_like(lw) if lw is not None else None
        dlb = torch.zeros_like(lb) if lb is not None else None
        g_direct, g_agg = gy, None
        gsum = None                                  # per-colour gradient sums entering the quotient stage below
        for k in range(2 * ctx.rounds - 1, -1, -1):
            rev = k % 2 == 1
            w = par[5:] if rev else par[:5]
            g = dict(acc['r' if rev else 'f'])
            g['dln_w'], g['dln_b'] = dlw, dlb
            if k >= len(quot):
                p, i = plan.csr(rev)
                g_direct, g_agg = struct_stage_bwd(ctx.states[k], p, i, xcls, w[0], w[1], w[2], w[3], w[4], lw, lb, g_direct, g_agg, g,
                                                   need_input_grad=(k > 0), wpack=ctx.packs[int(rev)], heavy=plan.heavy(rev), stats=ctx.stats[k])
                continue
            st = quot[k]
            if gsum is None:
                # the last quotient stage: per-colour sums of the per-node gradient (g_direct + the pull of g_agg over this stage's
                # lists), colour runs cut into segments, partial rows summed level by level (mgv_seg_sum: list order, no atomics)
                p, i = plan.csr(rev)
                order, levels = st['sum_levels']
                gsum = _seg_sums(H, levels, order, g_direct, g_agg, p, i)
            if use_x3(H):
                gd_c, ga_c = struct_stage_bwd(ctx.states[k], st['ptr'], st['ent_idx'], st['xcls'], w[0], w[1], w[2], w[3], w[4], lw, lb, gsum, None, g,
                                              need_input_grad=(k > 0), wpack=ctx.packs[int(rev)], heavy=st['heavy'], table_own=st['own32'], tagged=False,
                                              stats=ctx.stats[k])
            else:
                gd_c, ga_c = struct_stage_bwd(ctx.states[k], st['ptr'], st['idx'], st['xcls'], w[0], w[1], w[2], w[3], w[4], lw, lb, gsum, None, g,
                                              need_input_grad=(k > 0), wpack=ctx.packs[int(rev)], heavy=st['heavy'], n_rows=st['C'], stats=ctx.stats[k])
            if k > 0:
                # colour sums for stage k-1: a colour there collects the own-row gradients of the representatives that own it and the
                # aggregate gradients of those that list it (deterministic gathers over the colour-level lists)
                gsum = _seg_sums(H, st['own_levels'], st['own_rows'], gd_c) + _seg_sums(H, st['ent_levels'], st['ent_rows'], ga_c)
        ctx.states = ctx.stats = None
        f, r = acc['f'], acc['r']
        return (None, None, None, f['dxtab'], f['dWc'], f['dbc'], f['dWhh'], f['dbhh'],
                r['dxtab'], r['dWc'], r['dbc'], r['dWhh'], r['dbhh'], dlw, dlb)

    @staticmethod
    def backward(ctx, gy):
        plan, xcls, par, lw, lb = ctx.plan, ctx.xcls, ctx.par, ctx.lw, ctx.lb
        gy = gy.contiguous()
        if ctx.quot:
            return StructEncoderFn._backward_quotient(ctx, gy)
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
                if k == 0 and ctx.first is not None:
                    # parameter gradients are linear in the incoming gradient: sum it per (degree, class) pair,
                    # then one backward row per pair
                    cid, C, tp, ti, tx = ctx.first
                    H = g_direct.shape[1]
                    gsum = torch.zeros(C, H, dtype=F32, device=g_direct.device)
                    ws = workspace(_hip.call_value('mgv_class_pull_sum_ws_floats', H, plan.N, C), g_direct.device)
                    _hip.call('mgv_class_pull_sum', H, plan.N, ptr(g_direct), ptr(g_agg), ptr(p), ptr(i), ptr(cid), C, ptr(gsum), ptr(ws), ws.numel())
                    struct_stage_bwd(torch.ones(C, H, dtype=F32, device=g_direct.device), tp, ti, tx, w[0], w[1], w[2], w[3], w[4],
                                     lw, lb, gsum, None, g, need_input_grad=False, wpack=ctx.packs[0], stats=ctx.stats[k])
                elif isinstance(ctx.states[k], tuple):
                    g_direct, g_agg = struct_stage_bwd(ctx.states[k][1], p, plan.tagged_idx(rev, ctx.first[0]), xcls, w[0], w[1], w[2], w[3], w[4],
                                                       lw, lb, g_direct, g_agg, g, need_input_grad=(k > 0), wpack=ctx.packs[int(rev)],
                                                       heavy=plan.heavy(rev), table_own=ctx.first[0], stats=ctx.stats[k])
                else:
                    g_direct, g_agg = struct_stage_bwd(ctx.states[k], p, i, xcls, w[0], w[1], w[2], w[3], w[4], lw, lb,
                                                       g_direct, g_agg, g, need_input_grad=(k > 0), wpack=ctx.packs[int(rev)],
                                                       heavy=plan.heavy(rev), stats=ctx.stats[k])
                k -= 1
        ctx.states = ctx.stats = None
        f, r = acc['f'], acc['r']
        return (None, None, None, f['dxtab'], f['dWc'], f['dbc'], f['dWhh'], f['dbhh'],
                r['dxtab'], r['dWc'], r['dbc'], r['dWhh'], r['dbhh'], dlw, dlb)


class StructEncoderRowsFn(torch.autograd.Function):
    """MultiGCNEncoder (digae_layer.py:257-277) for GENERAL node features: x has more distinct rows than the class table holds, so
    the GRU's feature term enters per node — xrow_f / xrow_r [N, 3H] = x W_ih[:, H:]^T + b_ih of the forward / reversed half, formed
    by the caller with ops.linear (autograd carries their gradients to W_ih, b_ih and x through the linear kernels).  Every half
    round runs per node on the exact-fp32 stage kernels (mgv_struct_stage_rows_fwd / _bwd): rows differ node by node, so neither
    the (degree, class) table nor the colour quotient applies.  The reference's Models never reach this path (they feed one-hot
    rows, dg_ae_model_aig.py:59); it completes the encoder's surface."""

    @staticmethod
    def forward(ctx, plan, rounds, xrow_f, Wc_f, bc_f, Whh_f, bhh_f, xrow_r, Wc_r, bc_r, Whh_r, bhh_r, ln_w, ln_b):
        N, H, dev = plan.N, Whh_f.shape[1], Whh_f.device
        par = [check(t.detach().contiguous(), F32, 'encoder parameter') for t in (xrow_f, Wc_f, bc_f, Whh_f, bhh_f, xrow_r, Wc_r, bc_r, Whh_r, bhh_r)]
        lw = ln_w.detach().contiguous() if ln_w is not None else None
        lb = ln_b.detach().contiguous() if ln_b is not None else None
        assert par[0].shape == (N, 3 * H) and par[5].shape == (N, 3 * H)
        h = torch.ones(N, H, dtype=F32, device=dev)
        states = []
        for _ in range(rounds):
            for rev in (False, True):
                p, i = plan.csr(rev)
                w = par[5:] if rev else par[:5]
                states.append(h)
                out = torch.empty(N, H, dtype=F32, device=dev)
                _hip.call('mgv_struct_stage_rows_fwd', H, N, ptr(h), ptr(p), ptr(i), ptr(w[0]), ptr(w[1]), ptr(w[2]), ptr(w[3]), ptr(w[4]),
                          ptr(lw), ptr(lb), LN_EPS, ptr(out))
                h = out
        ctx.plan, ctx.rounds, ctx.par, ctx.lw, ctx.lb, ctx.states = plan, rounds, par, lw, lb, states
        return h

    @staticmethod
    def backward(ctx, gy):
        plan, par, lw, lb = ctx.plan, ctx.par, ctx.lw, ctx.lb
        N, H, dev = plan.N, gy.shape[1], gy.device
        acc = {}
        for tag, w in (('f', par[:5]), ('r', par[5:])):
            acc[tag] = [torch.zeros_like(t) for t in w]          # d_xrow, dWc, dbc, dWhh, dbhh
            acc[tag] += [w[1].t().contiguous(), w[3].t().contiguous()]
        dlw = torch.zeros_like(lw) if lw is not None else None
        dlb = torch.zeros_like(lb) if lb is not None else None
        g_direct, g_agg = gy.contiguous(), None
        k = len(ctx.states) - 1
        for _ in range(ctx.rounds):
            for rev in (True, False):
                p, i = plan.csr(rev)
                w = par[5:] if rev else par[:5]
                g = acc['r' if rev else 'f']
                gd = torch.empty(N, H, dtype=F32, device=dev) if k > 0 else None
                ga = torch.empty(N, H, dtype=F32, device=dev) if k > 0 else None
                _hip.call('mgv_struct_stage_rows_bwd', H, N, ptr(ctx.states[k]), ptr(p), ptr(i), ptr(w[0]), ptr(w[1]), ptr(g[5]), ptr(w[2]),
                          ptr(w[3]), ptr(g[6]), ptr(w[4]), ptr(lw), ptr(lb), LN_EPS, ptr(g_direct), ptr(g_agg), ptr(gd), ptr(ga),
                          ptr(g[1]), ptr(g[2]), ptr(g[3]), ptr(g[4]), ptr(g[0]), ptr(dlw), ptr(dlb))
                g_direct, g_agg = gd, ga
                k -= 1
        ctx.states = None
        f, r = acc['f'], acc['r']
        return (None, None, f[0], f[1], f[2], f[3], f[4], r[0], r[1], r[2], r[3], r[4], dlw, dlb)


# ------------------------------------------------------------------------------------------------
# Linear over node rows (hs_linear / hs_decompose / VAE heads / readout layers)
# ------------------------------------------------------------------------------------------------
_LIN_X3 = {}


def _lin_x3(M, K):
    """Layer shapes served by the bf16x3 linear kernels (the others stay on the fp32 MFMA ones)."""
    if PRECISION != 'x3':
        return False
    key = (int(M), int(K))
    if key not in _LIN_X3:
        _LIN_X3[key] = bool(_hip.call_value('mgv_linear_x3_supported', *key))
    return _LIN_X3[key]


def _lin_fwd(x1, x2, W, b, M, wpack=None, res=None):
    """wpack given: W is ignored (bf16x3 path with a ready fragment-order pack).  res [N, M]: added to the result (in the
    kernel's store phase on the bf16x3 path)."""
    N, K1 = x1.shape
    K2 = 0 if x2 is None else x2.shape[1]
    check(x1, F32, 'x1'); check(x2, F32, 'x2'); check(b, F32, 'b')
    y = torch.empty(N, M, dtype=F32, device=x1.device)
    if wpack is not None or _lin_x3(M, K1 + K2):
        if wpack is None:
            wpack = linear_wpack(check(W, F32, 'W'))
        if res is not None:
            res = _rowmajor(check(res, F32, 'res'))
            _hip.call('mgv_linear_fwd_x3_res', N, ptr(x1), K1, x1.stride(0), ptr(x2), K2, 0 if x2 is None else x2.stride(0),
                      ptr(wpack), ptr(b), M, ptr(res), res.stride(0), ptr(y), M)
        else:
            _hip.call('mgv_linear_fwd_x3', N, ptr(x1), K1, x1.stride(0), ptr(x2), K2, 0 if x2 is None else x2.stride(0),
                      ptr(wpack), ptr(b), M, ptr(y), M)
        return y
    check(W, F32, 'W')
    assert W.shape == (M, K1 + K2)
    _hip.call('mgv_linear_fwd', N, ptr(x1), K1, x1.stride(0), ptr(x2), K2, 0 if x2 is None else x2.stride(0),
              ptr(W), ptr(b), M, ptr(y), M)
    return y if res is None else y.add_(res)


def _rowmajor(t):
    """Tensors whose rows are contiguous (column slices of a wider matrix are fine)."""
    if t is None:
        return None
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0:
        return t
    return t.contiguous()


class LinearFn(torch.autograd.Function):
    """y = [x1 | x2] W^T + b on the MFMA row-streaming kernel (x2 optional: fuses torch.cat).
    `passthrough`: also return x1 itself; a second consumer of x1 that reads THIS output hands its gradient to this node's
    backward, whose input-gradient kernel adds it in its store phase — autograd then sees one consumer of x1, no N x K add."""

    @staticmethod
    def forward(ctx, x1, x2, W, b, passthrough=False):
        x1d, x2d = _rowmajor(x1.detach()), _rowmajor(x2.detach()) if x2 is not None else None
        Wd = W.detach().contiguous()
        bd = b.detach().contiguous() if b is not None else None
        ctx.save_for_backward(x1d, x2d, Wd)
        ctx.has_b = b is not None
        ctx.set_materialize_grads(False)
        y = _lin_fwd(x1d, x2d, Wd, bd, W.shape[0])
        return (y, x1.view_as(x1)) if passthrough else y

    @staticmethod
    def backward(ctx, gy, g_pass=None):
        x1, x2, W = ctx.saved_tensors
        if gy is None:                     # only the pass-through output was used
            return g_pass, None, None, None, None
        gy = _rowmajor(gy).contiguous()      # (a column slice of a wider gradient, e.g. behind torch.cat, arrives row-strided)
        N, K1 = x1.shape
        K2 = 0 if x2 is None else x2.shape[1]
        M = W.shape[0]
        gx1 = gx2 = gW = gb = None
        def dgrad(Ws, Kx, res=None):    # gy [N, M] x Ws [M, Kx]: the transposed weight view is packed straight from W
            if _lin_x3(Kx, M):
                pack = torch.empty(2 * Ws.numel(), dtype=torch.bfloat16, device=W.device)
                _pack_into(pack, 0, Ws, True)
                return _lin_fwd(gy, None, None, None, Kx, wpack=pack, res=res)
            return _lin_fwd(gy, None, Ws.t().contiguous(), None, Kx, res=res)
        if ctx.needs_input_grad[0]:
            gx1 = dgrad(W[:, :K1], K1, res=g_pass.detach() if g_pass is not None else None)
        if x2 is not None and ctx.needs_input_grad[1]:
            gx2 = dgrad(W[:, K1:], K2)
        if ctx.needs_input_grad[2] or (ctx.has_b and ctx.needs_input_grad[3]):
            gW = torch.zeros_like(W)
            gb = torch.zeros(M, dtype=F32, device=W.device) if ctx.has_b else None
            if _lin_x3(M, K1 + K2):
                ws = workspace(_hip.call_value('mgv_linear_wgrad_x3_ws_floats', M, K1 + K2, N), W.device)
                _hip.call('mgv_linear_wgrad_x3', N, ptr(x1), K1, x1.stride(0), ptr(x2), K2, 0 if x2 is None else x2.stride(0),
                          ptr(gy), gy.stride(0), M, ptr(gW), ptr(gb), ptr(ws), ws.numel())
            else:
                _hip.call('mgv_linear_wgrad', N, ptr(x1), K1, x1.stride(0), ptr(x2), K2, 0 if x2 is None else x2.stride(0),
                          ptr(gy), gy.stride(0), M, ptr(gW), ptr(gb))
        return gx1, gx2, gW, gb, None


def linear(x, W, b=None, x2=None):
    return LinearFn.apply(x, x2, W, b)


def linear_passthrough(x, W, b=None):
    """(y, x'): x' is x; feed x' to the other consumer of x and its gradient is added inside this Linear's input-gradient kernel."""
    return LinearFn.apply(x, None, W, b, True)


class RoundGhFn(torch.autograd.Function):
    """gh[N, 3H] = W_hh[slot(v)] h[v] + b_hh[slot(v)] for every node v the sweep updates (zero rows elsewhere): the hidden half of each
    gate's own GRU for rounds >= 2 (dg_ae_model_aig.py:70,88-94), as ONE grouped Linear over the sweep's (level, slot) tiles
    (csrc/linear_x3.hip grouped mode) — and its backward: the input gradient as a grouped Linear with the transposed packs, the weight
    and bias gradients per slot over the slot's tile list.  W [T, 3H, H], b [T, 3H] stacked in slot order."""

    @staticmethod
    def forward(ctx, plan, h, W, b):
        hd, Wd, bd = check(h.detach().contiguous(), F32, 'h'), check(W.detach().contiguous(), F32, 'W'), check(b.detach().contiguous(), F32, 'b')
        N, H = hd.shape
        T, M = Wd.shape[0], 3 * H
        assert plan.has_levels and plan.num_slots == T and plan.N == N and Wd.shape == (T, M, H) and bd.shape == (T, M)
        pack = torch.empty(T, 2, M * H, dtype=torch.bfloat16, device=hd.device)
        for s_ in range(T):
            _hip.call('mgv_wpack_bf16x3', ptr(Wd[s_]), M, H, H, 0, ptr(pack[s_, 0]), ptr(pack[s_, 1]))
        gh = torch.zeros(N, M, dtype=F32, device=hd.device)
        _hip.call('mgv_grouped_linear_fwd_x3', plan.num_tiles, None, ptr(plan.order), ptr(plan.tile_start), ptr(plan.tile_count), ptr(plan.tile_slot),
                  ptr(hd), H, H, ptr(pack), ptr(bd), M, None, 0, ptr(gh), M)
        ctx.save_for_backward(hd, Wd)
        ctx.plan = plan
        return gh

    @staticmethod
    def backward(ctx, dgh):
        hd, Wd = ctx.saved_tensors
        plan = ctx.plan
        d = check(dgh.contiguous(), F32, 'dgh')
        N, H = hd.shape
        T, M = Wd.shape[0], 3 * H
        packT = torch.empty(T, 2, H * M, dtype=torch.bfloat16, device=hd.device)
        for s_ in range(T):
            _hip.call('mgv_wpack_bf16x3', ptr(Wd[s_]), H, M, H, 1, ptr(packT[s_, 0]), ptr(packT[s_, 1]))      # the transposed view of W[s]
        dh = torch.zeros(N, H, dtype=F32, device=hd.device)
        _hip.call('mgv_grouped_linear_fwd_x3', plan.num_tiles, None, ptr(plan.order), ptr(plan.tile_start), ptr(plan.tile_count), ptr(plan.tile_slot),
                  ptr(d), M, M, ptr(packT), None, H, None, 0, ptr(dh), H)
        dW, db = torch.zeros_like(Wd), torch.zeros(T, M, dtype=F32, device=hd.device)
        stp = plan.slot_tile_ptr
        for s_ in range(T):
            n_t = stp[s_ + 1] - stp[s_]
            if n_t == 0:
                continue
            ws = workspace(_hip.call_value('mgv_grouped_linear_wgrad_x3_ws_floats', M, H, n_t), hd.device)
            tiles = plan.slot_tiles[stp[s_]:]
            _hip.call('mgv_grouped_linear_wgrad_x3', n_t, ptr(tiles), ptr(plan.order), ptr(plan.tile_start), ptr(plan.tile_count), ptr(hd), H, H,
                      ptr(d), M, M, ptr(dW[s_]), ptr(db[s_]), ptr(ws), ws.numel())
        return None, dh, dW, db


class GatherSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, plan, reverse):
        hd = h.detach().contiguous()
        N, H = hd.shape
        p, i = plan.csr(reverse)
        agg = torch.empty_like(hd)
        deg = torch.empty(N, dtype=F32, device=hd.device)
        _hip.call('mgv_gather_sum', H, N, ptr(hd), ptr(p), ptr(i), ptr(agg), ptr(deg))
        ctx.plan, ctx.reverse = plan, reverse
        ctx.mark_non_differentiable(deg)
        return agg, deg

    @staticmethod
    def backward(ctx, gagg, _gdeg):
        g = gagg.contiguous()
        N, H = g.shape
        p, i = ctx.plan.csr(not ctx.reverse)      # the scatter of a gather is the gather over the flipped CSR
        out = torch.empty_like(g)
        _hip.call('mgv_gather_sum', H, N, ptr(g), ptr(p), ptr(i), ptr(out), None)
        return out, None, None


def gather_sum(h, nbr_ptr=None, nbr_idx=None, plan=None, reverse=False):
    if plan is not None:
        return GatherSumFn.apply(h, plan, reverse)
    hd = check(h.detach().contiguous(), F32, 'h')
    N, H = hd.shape
    agg = torch.empty_like(hd)
    deg = torch.empty(N, dtype=F32, device=hd.device)
    _hip.call('mgv_gather_sum', H, N, ptr(hd), ptr(nbr_ptr), ptr(nbr_idx), ptr(agg), ptr(deg))
    return agg, deg


class AttnPoolFn(torch.autograd.Function):
    """zbar[i] = sum_j softmax_j(u . x_j) x_j over i's in-edges (csrc/attn_pool.hip): the stand-alone TFMlpAggr call."""

    @staticmethod
    def forward(ctx, x, u, plan):
        xd = check(x.detach().contiguous(), F32, 'x')
        ud = check(u.detach().contiguous(), F32, 'u')
        N, W = xd.shape
        if W not in (32, 64, 128) or ud.numel() != W:
            raise ValueError('attention pooling: row width %d (u: %d) is not one of 32 / 64 / 128' % (W, ud.numel()))
        zbar = torch.empty_like(xd)
        m, inv = torch.empty(N, dtype=F32, device=xd.device), torch.empty(N, dtype=F32, device=xd.device)
        _hip.call('mgv_attn_pool_fwd', W, N, ptr(plan.in_ptr), ptr(plan.in_src), ptr(xd), ptr(ud), ptr(zbar), ptr(m), ptr(inv))
        ctx.plan = plan
        ctx.save_for_backward(xd, ud, zbar, m, inv)
        return zbar

    @staticmethod
    def backward(ctx, gz):
        xd, ud, zbar, m, inv = ctx.saved_tensors
        N, W = xd.shape
        g = _rowmajor(gz).contiguous()
        dx, du = torch.zeros_like(xd), torch.zeros_like(ud)
        _hip.call('mgv_attn_pool_bwd', W, N, ptr(ctx.plan.in_ptr), ptr(ctx.plan.in_src), ptr(xd), ptr(ud), ptr(zbar), ptr(m), ptr(inv),
                  ptr(g), ptr(dx), ptr(du))
        return dx, du, None


# ------------------------------------------------------------------------------------------------
# levelised functional sweep
# ------------------------------------------------------------------------------------------------
_SWEEP_BWD_EVENT = None


def _sweep_bwd_prep(plan, T, H, dev):
    """Scratch and heavy-list arguments of mgv_func_sweep_bwd_x3 / mgv_func_sweep_round_bwd_x3."""
    ltp = plan.level_tile_ptr
    widest = max([ltp[i + 1] - ltp[i] for i in range(1, len(ltp) - 1)] + [1])
    # rows for the deferred weight gradient, small-gradient slabs of the widest level, 256 rows of weight-gradient partials
    scratch = torch.empty(plan.n_active * 5 * H + widest * T * 11 * H + 256 * 6 * H * H, dtype=F32, device=dev)
    stp = (_hip.ctypes.c_int32 * len(plan.slot_tile_ptr))(*plan.slot_tile_ptr)
    hv = plan.heavy_segments(True, inactive_only=True)
    hav = plan.heavy_segments(True, active_by_level=True)      # updated gates with very long consumer lists
    if hav is None:
        ha = (0, None, None, None, None, None, None, None, 0)
    else:
        i32a = _hip.ctypes.c_int32
        kp = (i32a * len(hav['lvl_k_ptr']))(*hav['lvl_k_ptr'])
        sp_ = (i32a * len(hav['lvl_seg_ptr']))(*hav['lvl_seg_ptr'])
        need = (hav['K'] + hav['S']) * 2 * H          # its own buffer: it must outlive the launcher call's other scratch users
        hws = _WS.get(('heavy_active', str(dev)))
        if hws is None or hws.numel() < need:
            hws = _WS[('heavy_active', str(dev))] = torch.empty(need, dtype=F32, device=dev)
        ha = (hav['K'], ptr(hav['nodes']), ptr(hav['node_seg_ptr']), ptr(hav['seg_e0']), ptr(hav['seg_e1']), kp, sp_, ptr(hws), plan.HEAVY_ROW)
    return scratch, stp, hv, ha


def _sweep_x3(H):
    """The bf16x3 level kernels serve this width (else the exact-fp32 ones: H = 16, MGV_PRECISION=f32, MGV_SWEEP_X3=0)."""
    return use_x3(H) and os.environ.get('MGV_SWEEP_X3', '1') != '0'


# The sweep as ONE persistent kernel per direction (csrc/sweep_persist_x3.hip: slot-dedicated workgroups, weights resident in LDS, XCD
# grid barrier, in-register weight gradient).  Built, parity-tested and MEASURED in round 4: not faster than the per-level kernels at any
# batch size (config 2: forward 2.17 vs 2.00 ms, backward 8.2 vs 7.4 ms; one graph: 0.95 / 2.80 vs 1.01 / 2.84 ms), so it is
# opt-in (MGV_SWEEP_PERSIST=1) and the per-level kernels stay the default.
GROUPED_ROUND = True            # rounds >= 2: W_hh h_prev + b_hh of every updated gate as one grouped Linear (RoundGhFn); False: per gate type on the plain kernels

# The level kernels read packed rows (GraphPlan.order_rows: spans + first in-edge sources + first consumers, one 128-byte line per
# updated node) or the 16-byte span rows and the CSR lists behind them.  Packed rows cost 0.43 ms to build and save 0.13 ms per sweep
# backward (config 2): they pay from a plan's THIRD step on, so a plan gets them when it comes back for a second step (a resident
# batch) and a batch that is planned, stepped once and dropped (every batch of a shuffled training loop) never builds them.
# MGV_PACKED_ROWS=0: never; =2: from the first step.
PACKED_ROWS = {'0': 0, '2': 2}.get(os.environ.get('MGV_PACKED_ROWS', '1'), 1)


def _sweep_rows(plan, forward=False):
    """(pointer, ints per row) of the rows the level kernels read; `forward`: a sweep forward (counts the plan's steps)."""
    if PACKED_ROWS and forward and plan.__dict__.get('_order_rows') is None:
        steps = plan.__dict__.get('_sweep_steps', 0)
        plan._sweep_steps = steps + 1
        if PACKED_ROWS == 2 or steps >= 1:
            plan.order_rows                  # (built here, on the step's stream: 0.43 ms once; the first step's backward keeps the span rows)
    if PACKED_ROWS and plan.__dict__.get('_order_rows') is not None:
        return ptr(plan.order_rows), 32
    return ptr(plan.order_span), 4


PERSIST = os.environ.get('MGV_SWEEP_PERSIST', '0') == '1'


def _persist_ws(dev):
    """(barrier state, sticky status word) of the persistent sweep kernels on `dev`: the launchers zero the first before every
    launch, the second is zeroed here once and only ever set by a kernel whose barrier gave up (persist_check)."""
    key = ('persist', str(dev))
    ws = _WS.get(key)
    if ws is None:
        nb = _hip.call_value('mgv_sweep_persist_sync_bytes')
        ws = _WS[key] = (torch.zeros((nb + 3) // 4, dtype=torch.int32, device=dev), torch.zeros(4, dtype=torch.int32, device=dev),
                         _hip.call_value('mgv_sweep_persist_max_grid'))
    return ws


def persist_status(dev):
    """The sticky status word as a device tensor (one int32; the trainer copies it back with the step metrics)."""
    return _persist_ws(dev)[1][:1]


def persist_check(dev=None):
    """Raise if a persistent sweep kernel on `dev` ever gave up at its grid barrier (synchronises)."""
    for key, ws in list(_WS.items()):
        if key[0] == 'persist' and (dev is None or key[1] == str(dev)):
            code = int(ws[1][0].item())
            if code != 0:
                raise HipLibraryError('persistent sweep kernel gave up at its grid barrier (code %d): its results are invalid' % code)


def _persist_roles(plan, H, N):
    """wg_begin (ctypes array) when the persistent sweep kernels serve this plan, else None (-> the per-level kernels)."""
    if not (PERSIST and H == 64 and plan.num_levels > 2 and N * 2 * H * 4 < (1 << 32) and getattr(plan, 'key_tile_ptr', None) is not None):
        return None
    roles = plan.persist_roles(_persist_ws(plan.device)[2])
    if roles is None:
        return None
    return (_hip.ctypes.c_int32 * len(roles))(*roles)


class FuncSweepRoundFn(torch.autograd.Function):
    """Round r >= 2 of the functional sweep (dg_ae_model_aig.py:70-97 with num_rounds > 1) on the HIP level kernels (bf16x3 or
    exact fp32, as round 1): hf_new = sweep(hs, hf_prev) where every updated gate's GRU starts from its previous state.
    `gh` [N, 3H] = W_hh h_prev + b_hh of each node's own aggregator, formed by the caller with ops.linear (so that autograd carries
    its gradient to W_hh, b_hh and h_prev through the linear kernels); the level kernels add it to the gate pre-activations, mix
    z * h_prev into the new state and leave d(gh) and dh * z on the way back.  High fan-out lists take the same pre-passes as in
    round 1 (GraphPlan.heavy_segments)."""

    @staticmethod
    def forward(ctx, plan, hs, hprev, gh, attn_u, Wvc, bvc, bih):
        hsd = check(hs.detach().contiguous(), F32, 'hs')
        hp = check(hprev.detach().contiguous(), F32, 'h_prev')
        ghd = check(gh.detach().contiguous(), F32, 'gh')
        N, H = hsd.shape
        par = [check(t.detach().contiguous(), F32, 'sweep parameter') for t in (attn_u, Wvc, bvc, bih)]
        T = par[0].shape[0]
        assert plan.has_levels and plan.num_slots == T and plan.N == N and ghd.shape == (N, 3 * H) and hp.shape == (N, H)
        ltp = (_hip.ctypes.c_int32 * len(plan.level_tile_ptr))(*plan.level_tile_ptr)
        wpack = sweep_wpack(par[1]) if _sweep_x3(H) else None
        zb = torch.zeros(T, 3 * H, dtype=F32, device=hsd.device)
        hf = hp.clone()                      # never-updated rows keep their state; every updated row is rewritten by its level
        if wpack is not None:
            _hip.call('mgv_func_sweep_round_fwd_x3', H, N, T, plan.num_levels, ltp, ptr(plan.order), *_sweep_rows(plan, True),
                      ptr(plan.tile_start), ptr(plan.tile_count), ptr(plan.tile_slot), ptr(plan.in_ptr), ptr(plan.in_src), ptr(hsd), ptr(hf),
                      ptr(par[0]), ptr(wpack), ptr(par[2]), ptr(par[3]), ptr(zb), ptr(ghd), ptr(hp))
        else:
            _hip.call('mgv_func_sweep_round_fwd', H, N, T, plan.num_levels, ltp, ptr(plan.order), ptr(plan.tile_start),
                      ptr(plan.tile_count), ptr(plan.tile_slot), ptr(plan.in_ptr), ptr(plan.in_src), ptr(hsd), ptr(hf),
                      ptr(par[0]), ptr(par[1]), ptr(par[2]), ptr(par[3]), ptr(zb), ptr(ghd), ptr(hp))
        ctx.plan, ctx.par, ctx.ltp, ctx.wpack, ctx.zb = plan, par, ltp, wpack, zb
        ctx.save_for_backward(hsd, hf, hp, ghd)
        return hf

    @staticmethod
    def backward(ctx, ghf):
        plan, par = ctx.plan, ctx.par
        hs, hf, hp, ghd = ctx.saved_tensors
        N, H = hs.shape
        T = par[0].shape[0]
        dev = hs.device
        ghf = check(ghf.contiguous(), F32, 'ghf')
        dzb = torch.empty(N, 2 * H, dtype=F32, device=dev)
        alpha = torch.empty(max(plan.E, 1), dtype=F32, device=dev)
        dsc = torch.empty(max(plan.E, 1), dtype=F32, device=dev)
        grads = [torch.zeros_like(t) for t in par] + [torch.zeros_like(ctx.zb)]      # the last one (dbhh) is not meaningful here
        d_gh = torch.zeros(N, 3 * H, dtype=F32, device=dev)          # rows of never-updated nodes stay zero
        g_hprev = torch.zeros(N, H, dtype=F32, device=dev)           # (their states are constants of round 1: no gradient to carry)
        if ctx.wpack is None:
            ghs = torch.zeros(N, H, dtype=F32, device=dev)           # the fp32 kernels add to it
            WvcT = par[1].transpose(1, 2).contiguous()
            _hip.call('mgv_func_sweep_round_bwd', H, N, T, plan.num_levels, ctx.ltp, ptr(plan.order), ptr(plan.tile_start),
                      ptr(plan.tile_count), ptr(plan.tile_slot), ptr(plan.in_ptr), ptr(plan.in_src), ptr(plan.out_ptr),
                      ptr(plan.out_dst), ptr(plan.out_slot), ptr(plan.gslot), ptr(hs), ptr(hf), ptr(par[0]), ptr(par[1]),
                      ptr(WvcT), ptr(par[2]), ptr(par[3]), ptr(ctx.zb), ptr(ghf), ptr(ghs), ptr(dzb), ptr(alpha), ptr(dsc),
                      *[ptr(g) for g in grads], ptr(ghd), ptr(hp), ptr(d_gh), ptr(g_hprev))
            return (None, ghs, g_hprev, d_gh, grads[0], grads[1], grads[2], grads[3])
        ghs = torch.empty(N, H, dtype=F32, device=dev)
        scratch, stp, hv, ha = _sweep_bwd_prep(plan, T, H, dev)
        _hip.call('mgv_func_sweep_round_bwd_x3', H, N, T, plan.num_levels, ctx.ltp, ptr(plan.order), *_sweep_rows(plan),
                  plan.n_active, ptr(plan.tile_start), ptr(plan.tile_count), ptr(plan.tile_slot), ptr(plan.slot_tiles), stp,
                  ptr(plan.in_ptr), ptr(plan.in_src), ptr(plan.out_ptr), ptr(plan.out_dst), ptr(plan.out_slot),
                  ptr(plan.gslot), ptr(hs), ptr(hf), ptr(par[0]), ptr(ctx.wpack), ptr(par[2]), ptr(par[3]), ptr(ctx.zb),
                  ptr(ghf), ptr(ghs), ptr(dzb), ptr(alpha), ptr(dsc), *[ptr(g) for g in grads], ptr(scratch),
                  scratch.numel(), plan.HEAVY_ROW if hv is not None else 0, *ha, ptr(ghd), ptr(hp), ptr(d_gh), ptr(g_hprev))
        if hv is not None:
            # primary inputs (never updated) that drive thousands of gates: their pull by whole workgroups, per list segment
            pw = workspace(hv['S'] * H, dev)
            _hip.call('mgv_sweep_pull_heavy', H, hv['K'], ptr(hv['nodes']), ptr(hv['node_seg_ptr']), hv['S'], ptr(hv['seg_e0']), ptr(hv['seg_e1']),
                      ptr(plan.out_dst), ptr(plan.out_slot), ptr(plan.gslot), ptr(alpha), ptr(dsc), ptr(dzb), ptr(par[0]), ptr(pw), ptr(ghs))
        return (None, ghs, g_hprev, d_gh, grads[0], grads[1], grads[2], grads[3])


class FuncSweepFn(torch.autograd.Function):
    """hf = sweep(hs) over levels 1..L-1 (dg_ae_model_aig.py:70-97); parameters are the per-slot
    composed tensors attn_u [T,2H], Wvc [T,3H,2H], bvc/bih/bhh [T,3H]."""

    @staticmethod
    def forward(ctx, plan, hs, attn_u, Wvc, bvc, bih, bhh):
        hsd = check(hs.detach().contiguous(), F32, 'hs')
        N, H = hsd.shape
        par = [check(t.detach().contiguous(), F32, 'sweep parameter') for t in (attn_u, Wvc, bvc, bih, bhh)]
        T = par[0].shape[0]
        assert plan.has_levels and plan.num_slots == T and plan.N == N
        ltp = (_hip.ctypes.c_int32 * len(plan.level_tile_ptr))(*plan.level_tile_ptr)
        wpack = sweep_wpack(par[1]) if _sweep_x3(H) else None
        if wpack is not None:
            hf = torch.empty(N, H, dtype=F32, device=hsd.device)          # every updated row is written by its level; the rest here
            _hip.call('mgv_sweep_zero_inactive', H, N, ptr(plan.gslot), ptr(hf))
        else:
            hf = torch.zeros(N, H, dtype=F32, device=hsd.device)
        roles = _persist_roles(plan, H, N) if wpack is not None else None
        if roles is not None:
            sync, sticky, _ = _persist_ws(hsd.device)
            stp = (_hip.ctypes.c_int32 * len(plan.slot_tile_ptr))(*plan.slot_tile_ptr)
            _hip.call('mgv_func_sweep_fwd_persist_x3', H, N, T, plan.num_levels, ptr(plan.key_tile_ptr), roles, stp, ptr(plan.order),
                      ptr(plan.order_span), ptr(plan.tile_start), ptr(plan.tile_count), ptr(plan.in_ptr), ptr(plan.in_src), ptr(hsd),
                      ptr(hf), ptr(par[0]), ptr(wpack), ptr(par[2]), ptr(par[3]), ptr(par[4]), ptr(sync), ptr(sticky))
        elif wpack is not None:
            _hip.call('mgv_func_sweep_fwd_x3', H, N, T, plan.num_levels, ltp, ptr(plan.order), *_sweep_rows(plan, True),
                      ptr(plan.tile_start), ptr(plan.tile_count), ptr(plan.tile_slot), ptr(plan.in_ptr), ptr(plan.in_src), ptr(hsd), ptr(hf),
                      ptr(par[0]), ptr(wpack), ptr(par[2]), ptr(par[3]), ptr(par[4]))
        else:
            _hip.call('mgv_func_sweep_fwd', H, N, T, plan.num_levels, ltp, ptr(plan.order), ptr(plan.tile_start),
                      ptr(plan.tile_count), ptr(plan.tile_slot), ptr(plan.in_ptr), ptr(plan.in_src), ptr(hsd), ptr(hf),
                      *[ptr(t) for t in par])
        ctx.plan, ctx.par, ctx.ltp, ctx.wpack = plan, par, ltp, wpack
        ctx.save_for_backward(hsd, hf)
        return hf

    @staticmethod
    def backward(ctx, ghf):
        # start signal for work that should run BESIDE the (latency-bound) sweep backward on another stream: the reconstruction
        # branch's backward waits for it (ReconLossFn.backward), instead of starting beside the bandwidth-bound readout backward
        global _SWEEP_BWD_EVENT
        if ghf.is_cuda:
            _SWEEP_BWD_EVENT = torch.cuda.Event()
            _SWEEP_BWD_EVENT.record()
        plan, par = ctx.plan, ctx.par
        hs, hf = ctx.saved_tensors
        N, H = hs.shape
        T = par[0].shape[0]
        dev = hs.device
        ghf = check(ghf.contiguous(), F32, 'ghf')
        ghs = (torch.empty if ctx.wpack is not None else torch.zeros)(N, H, dtype=F32, device=dev)
        dzb = torch.empty(N, 2 * H, dtype=F32, device=dev)
        alpha = torch.empty(max(plan.E, 1), dtype=F32, device=dev)
        dsc = torch.empty(max(plan.E, 1), dtype=F32, device=dev)
        grads = [torch.zeros_like(t) for t in par]
        roles = _persist_roles(plan, H, N) if ctx.wpack is not None else None
        if roles is not None and plan.heavy_segments(True, active_by_level=True) is None:
            sync, sticky, _ = _persist_ws(dev)
            stp = (_hip.ctypes.c_int32 * len(plan.slot_tile_ptr))(*plan.slot_tile_ptr)
            hv = plan.heavy_segments(True, inactive_only=True)
            slab = workspace(_hip.call_value('mgv_sweep_persist_slab_floats', H, roles[T]), dev)
            _hip.call('mgv_func_sweep_bwd_persist_x3', H, N, T, plan.num_levels, ptr(plan.key_tile_ptr), roles, stp, ptr(plan.order),
                      ptr(plan.order_span), ptr(plan.tile_start), ptr(plan.tile_count), ptr(plan.in_ptr), ptr(plan.in_src),
                      ptr(plan.out_ptr), ptr(plan.out_dst), ptr(plan.out_slot), ptr(plan.gslot), ptr(hs), ptr(hf), ptr(par[0]),
                      ptr(ctx.wpack), ptr(par[2]), ptr(par[3]), ptr(par[4]), ptr(ghf), ptr(ghs), ptr(dzb), ptr(alpha), ptr(dsc),
                      *[ptr(g) for g in grads], ptr(slab), slab.numel(), plan.HEAVY_ROW if hv is not None else 0, ptr(sync), ptr(sticky))
            if hv is not None:
                pw = workspace(hv['S'] * H, dev)
                _hip.call('mgv_sweep_pull_heavy', H, hv['K'], ptr(hv['nodes']), ptr(hv['node_seg_ptr']), hv['S'], ptr(hv['seg_e0']), ptr(hv['seg_e1']),
                          ptr(plan.out_dst), ptr(plan.out_slot), ptr(plan.gslot), ptr(alpha), ptr(dsc), ptr(dzb), ptr(par[0]), ptr(pw), ptr(ghs))
            return (None, ghs, *grads)
        if ctx.wpack is not None:
            scratch, stp, hv, ha = _sweep_bwd_prep(plan, T, H, dev)
            _hip.call('mgv_func_sweep_bwd_x3', H, N, T, plan.num_levels, ctx.ltp, ptr(plan.order), *_sweep_rows(plan),
                      plan.n_active, ptr(plan.tile_start), ptr(plan.tile_count), ptr(plan.tile_slot), ptr(plan.slot_tiles), stp,
                      ptr(plan.in_ptr), ptr(plan.in_src), ptr(plan.out_ptr), ptr(plan.out_dst), ptr(plan.out_slot),
                      ptr(plan.gslot), ptr(hs), ptr(hf), ptr(par[0]), ptr(ctx.wpack), ptr(par[2]), ptr(par[3]), ptr(par[4]),
                      ptr(ghf), ptr(ghs), ptr(dzb), ptr(alpha), ptr(dsc), *[ptr(g) for g in grads], ptr(scratch),
                      scratch.numel(), plan.HEAVY_ROW if hv is not None else 0, *ha)
            if hv is not None:
                # primary inputs (never updated) that drive thousands of gates: their pull by whole workgroups, per list segment
                pw = workspace(hv['S'] * H, dev)
                _hip.call('mgv_sweep_pull_heavy', H, hv['K'], ptr(hv['nodes']), ptr(hv['node_seg_ptr']), hv['S'], ptr(hv['seg_e0']), ptr(hv['seg_e1']),
                          ptr(plan.out_dst), ptr(plan.out_slot), ptr(plan.gslot), ptr(alpha), ptr(dsc), ptr(dzb), ptr(par[0]), ptr(pw), ptr(ghs))
            return (None, ghs, *grads)
        WvcT = par[1].transpose(1, 2).contiguous()
        _hip.call('mgv_func_sweep_bwd', H, N, T, plan.num_levels, ctx.ltp, ptr(plan.order), ptr(plan.tile_start),
                  ptr(plan.tile_count), ptr(plan.tile_slot), ptr(plan.in_ptr), ptr(plan.in_src), ptr(plan.out_ptr),
                  ptr(plan.out_dst), ptr(plan.out_slot), ptr(plan.gslot), ptr(hs), ptr(hf), ptr(par[0]), ptr(par[1]),
                  ptr(WvcT), ptr(par[2]), ptr(par[3]), ptr(par[4]), ptr(ghf), ptr(ghs), ptr(dzb), ptr(alpha), ptr(dsc),
                  *[ptr(g) for g in grads])
        return (None, ghs, *grads)


# ------------------------------------------------------------------------------------------------
# decoder, reconstruction loss, confusion counters
# ------------------------------------------------------------------------------------------------
def _edge_rows(edge_index):
    ei = edge_index
    if ei.dtype != torch.int64:
        ei = ei.long()
    return ei[0].contiguous(), ei[1].contiguous()


class EdgeDotFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s, t, edge_index, sigmoid):
        sd, td = check(s.detach().contiguous(), F32, 's'), check(t.detach().contiguous(), F32, 't')
        src, dst = _edge_rows(edge_index)
        E, H = src.numel(), sd.shape[1]
        out = torch.empty(E, dtype=F32, device=sd.device)
        _hip.call('mgv_edge_dot_fwd', H, E, ptr(sd), ptr(td), H, ptr(src), ptr(dst), int(bool(sigmoid)), ptr(out))
        ctx.save_for_backward(sd, td, src, dst)
        ctx.sigmoid = bool(sigmoid)
        return out

    @staticmethod
    def backward(ctx, gout):
        sd, td, src, dst = ctx.saved_tensors
        H = sd.shape[1]
        ds, dt = torch.zeros_like(sd), torch.zeros_like(td)
        gout = gout.contiguous()             # (named: the launcher takes a raw pointer)
        _hip.call('mgv_edge_dot_bwd', H, src.numel(), ptr(sd), ptr(td), H, ptr(src), ptr(dst), int(ctx.sigmoid),
                  ptr(gout), ptr(ds), ptr(dt))
        return ds, dt, None, None


def edge_dot(s, t, edge_index, sigmoid=True):
    return EdgeDotFn.apply(s, t, edge_index, sigmoid)


def dense_scores(s, t):
    """s t^T for `forward_all` (tiny graphs only): rows of t act as the weight of a Linear."""
    M = t.shape[0]
    if M not in (16, 32, 64, 128):
        raise NotImplementedError('forward_all is only provided for 16/32/64/128 target rows')
    return linear(s, t.contiguous(), None)


class ReconLossFn(torch.autograd.Function):
    """-mean log(sigma(<s_u,t_v>)+1e-15) over positives - mean log(1-sigma+1e-15) over negatives
    (dg_ae_model_aig.py:108-130) on st = hs_decompose(hs) [N,2H]; also the confusion counters and,
    on request, pred_bin."""

    @staticmethod
    def forward(ctx, st, pos_edge_index, neg_edge_index, want_pred, plan=None, neg_csr=None):
        """`plan`: GraphPlan whose edges are exactly pos_edge_index (any order) -> atomic-free positive half;
        `neg_csr`: (out_ptr, out_dst, in_ptr, in_src) over neg_edge_index (sampling.NegativeEdges) -> atomic-free
        negative half as well."""
        std = check(st.detach().contiguous(), F32, 'st')
        N, H2 = std.shape
        H = H2 // 2
        ps, pd = _edge_rows(pos_edge_index)
        ns, nd = _edge_rows(neg_edge_index)
        Ep, En = ps.numel(), ns.numel()
        dev = std.device
        sums = torch.zeros(2, dtype=torch.float64, device=dev)
        counts = torch.zeros(4, dtype=torch.int64, device=dev)
        pred = torch.empty(Ep + En, dtype=torch.int32, device=dev) if want_pred else None
        t_view = std[:, H:]
        _hip.call('mgv_recon_loss_fwd', H, ptr(std), ptr(t_view), H2, ptr(ps), ptr(pd), Ep, ptr(ns), ptr(nd), En,
                  ptr(sums), ptr(counts), ptr(pred), *_sw(std.device))
        loss = (sums[0] / max(Ep, 1) + sums[1] / max(En, 1)).to(F32)
        ctx.save_for_backward(std, ps, pd, ns, nd)
        ctx.plan, ctx.neg_csr = plan, neg_csr
        ctx.mark_non_differentiable(counts)
        if pred is not None:
            ctx.mark_non_differentiable(pred)
            return loss, counts, pred
        return loss, counts, torch.empty(0, dtype=torch.int32, device=dev)

    @staticmethod
    def backward(ctx, gloss, _gc, _gp):
        global _SWEEP_BWD_EVENT
        if _SWEEP_BWD_EVENT is not None and gloss.is_cuda:
            torch.cuda.current_stream().wait_event(_SWEEP_BWD_EVENT)
            _SWEEP_BWD_EVENT = None
        std, ps, pd, ns, nd = ctx.saved_tensors
        H2 = std.shape[1]
        H = H2 // 2
        g = gloss.detach().to(F32).reshape(1).contiguous()
        pl = ctx.plan
        if pl is not None and ctx.neg_csr is not None:
            dst_ = torch.empty_like(std)
            # positive lists of high fan-out / fan-in nodes: skipped by the per-node pull, summed per segment by whole workgroups
            heavy = [(0, pl.heavy_segments(True), pl.out_dst), (1, pl.heavy_segments(False), pl.in_src)]
            skip = pl.HEAVY_ROW if any(hv is not None for _, hv, _ in heavy) else 0
            _hip.call('mgv_recon_loss_bwd_csr', H, std.shape[0], ptr(std), ptr(std[:, H:]), H2, ptr(pl.out_ptr), ptr(pl.out_dst),
                      ptr(pl.in_ptr), ptr(pl.in_src), ps.numel(), *[ptr(c) for c in ctx.neg_csr], ns.numel(), ptr(g), ptr(dst_),
                      ptr(dst_[:, H:]), skip)
            for which, hv, lst in heavy:
                if hv is not None:
                    pw = workspace(hv['S'] * H, std.device)
                    _hip.call('mgv_recon_heavy_lists', H, ptr(std), ptr(std[:, H:]), H2, ps.numel(), ptr(g), hv['K'], ptr(hv['nodes']),
                              ptr(hv['node_seg_ptr']), hv['S'], ptr(hv['seg_node']), ptr(hv['seg_e0']), ptr(hv['seg_e1']), ptr(lst), which,
                              ptr(pw), ptr(dst_[:, H:]) if which else ptr(dst_))
            return dst_, None, None, None, None, None
        dst_ = torch.zeros_like(std)
        csr = (pl.out_ptr, pl.out_dst, pl.in_ptr, pl.in_src) if pl is not None else (None, None, None, None)
        _hip.call('mgv_recon_loss_bwd', H, std.shape[0], ptr(std), ptr(std[:, H:]), H2, ptr(ps), ptr(pd), ps.numel(),
                  *[ptr(c) for c in csr], ptr(ns), ptr(nd), ns.numel(), ptr(g), ptr(dst_), ptr(dst_[:, H:]))
        return dst_, None, None, None, None, None


def confusion_counts(pred_bin, gt_bin):
    """{TP, FP, TN, FN} counts (trainer.py:240-244) as an int64[4] device tensor."""
    counts = torch.zeros(4, dtype=torch.int64, device=pred_bin.device)
    _hip.call('mgv_confusion', pred_bin.numel(), ptr(check(pred_bin.contiguous(), torch.int32, 'pred_bin')),
              ptr(check(gt_bin.contiguous(), torch.int32, 'gt_bin')), ptr(counts))
    return counts


# ------------------------------------------------------------------------------------------------
# losses on node rows
# ------------------------------------------------------------------------------------------------
class L1LossFn(torch.autograd.Function):
    """nn.L1Loss() (mean) as used for the probability task (trainer.py:71,156)."""

    @staticmethod
    def forward(ctx, x, target):
        xd = check(x.detach().contiguous(), F32, 'x')
        td = check(target.detach().contiguous(), F32, 'target')
        assert xd.numel() == td.numel()
        s = torch.zeros(1, dtype=torch.float64, device=xd.device)
        _hip.call('mgv_l1_loss_fwd', xd.numel(), ptr(xd), ptr(td), ptr(s), *_sw(xd.device))
        ctx.save_for_backward(xd, td)
        return (s[0] / max(xd.numel(), 1)).to(F32)

    @staticmethod
    def backward(ctx, g):
        xd, td = ctx.saved_tensors
        dx = torch.empty_like(xd)
        gs = g.detach().to(F32).reshape(1).contiguous()
        _hip.call('mgv_l1_loss_bwd', xd.numel(), ptr(xd), ptr(td), ptr(gs), ptr(dx))
        return dx, None


def l1_loss(x, target):
    return L1LossFn.apply(x, target)


def pair_lists(tt_pair_index, num_nodes):
    """The truth-table pairs grouped by first and by second member: (a_ptr[N+1], a_pair[P], b_ptr[N+1], b_pair[P]), pair ids in
    their original order inside a group (csrc/plan_build.hip).  Static per batch: the trainer caches it on the batch."""
    pa, pb = _edge_rows(tt_pair_index)
    P, N, dev = pa.numel(), int(num_nodes), pa.device
    i32 = dict(dtype=I32, device=dev)
    a_ptr, b_ptr = torch.empty(N + 1, **i32), torch.empty(N + 1, **i32)
    junk = torch.empty(4, max(P, 1), **i32)                      # neighbour / slot arrays of the CSR build, not needed here
    a_pair, b_pair = torch.empty(max(P, 1), **i32), torch.empty(max(P, 1), **i32)
    n_s = _hip.call_value('mgv_plan_csr_scratch_ints', N, P)
    scratch = torch.empty(n_s, **i32)
    status = torch.empty(2, **i32)
    _hip.call('mgv_plan_csr', N, P, ptr(pa), ptr(pb), ptr(b_ptr), ptr(junk[0]), ptr(junk[1]), ptr(a_ptr), ptr(junk[2]), ptr(junk[3]),
              ptr(b_pair), ptr(a_pair), ptr(scratch), n_s, ptr(status))
    if int(status[0].item()) != 0:           # built once per batch and cached: one host read
        raise ValueError('tt_pair_index holds node ids outside [0, num_nodes)')
    return a_ptr, a_pair, b_ptr, b_pair


class FuncLossFn(torch.autograd.Function):
    """L1(z(1 - cos(hf[a], hf[b])), z(tt_sim)) with z = zero_normalization (trainer.py:158-163).
    `lists` (optional, from pair_lists): backward without atomics and without a zero-filled gradient.
    `passthrough`: also return hf itself; a second consumer of hf (the readout) that reads THIS output hands its gradient to
    this node's backward, whose pull kernel adds it on the way out — autograd then sees one consumer of hf and no N x H add."""

    @staticmethod
    def forward(ctx, hf, tt_pair_index, tt_sim, lists=None, passthrough=False):
        hfd = check(hf.detach().contiguous(), F32, 'hf')
        pa, pb = _edge_rows(tt_pair_index)
        tt = check(tt_sim.detach().to(F32).contiguous(), F32, 'tt_sim')
        P, H = pa.numel(), hfd.shape[1]
        dis = torch.empty(P, dtype=F32, device=hfd.device)
        ws = torch.zeros(8, dtype=torch.float64, device=hfd.device)
        _hip.call('mgv_func_loss_fwd', H, P, ptr(hfd), ptr(pa), ptr(pb), ptr(tt), 1e-8, ptr(dis), ptr(ws), *_sw(hfd.device))
        ctx.save_for_backward(hfd, pa, pb, tt, dis, ws)
        ctx.lists = lists
        ctx.set_materialize_grads(False)
        loss = (ws[4] / P).to(F32)
        if passthrough:
            return loss, hf.view_as(hf)
        return loss

    @staticmethod
    def backward(ctx, g, g_pass=None):
        hfd, pa, pb, tt, dis, ws = ctx.saved_tensors
        if g is None:                      # only the pass-through output was used
            return g_pass, None, None, None, None
        gs = g.detach().to(F32).reshape(1).contiguous()
        if ctx.lists is not None:
            dhf = torch.empty_like(hfd)
            add = check(g_pass.detach().contiguous(), F32, 'g_pass') if g_pass is not None else None
            _hip.call('mgv_func_loss_bwd_csr', hfd.shape[1], hfd.shape[0], pa.numel(), ptr(hfd), ptr(pa), ptr(pb), ptr(tt), ptr(dis), 1e-8,
                      ptr(ws), ptr(gs), *[ptr(t) for t in ctx.lists], ptr(add), ptr(dhf))
            return dhf, None, None, None, None
        dhf = torch.zeros_like(hfd) if g_pass is None else g_pass.detach().to(F32).clone()
        _hip.call('mgv_func_loss_bwd', hfd.shape[1], pa.numel(), ptr(hfd), ptr(pa), ptr(pb), ptr(tt), ptr(dis), 1e-8,
                  ptr(ws), ptr(gs), ptr(dhf))
        return dhf, None, None, None, None


def _pair_lists_for(hf, tt_pair_index, cache):
    lists = None
    if cache is not None and hf.is_cuda and tt_pair_index.shape[1] >= 2:
        lists = getattr(cache, '_mgv_pair_lists', None)
        if lists is None or lists[0].numel() != hf.shape[0] + 1 or lists[1].device != hf.device:
            lists = pair_lists(tt_pair_index, hf.shape[0])
            cache._mgv_pair_lists = lists
    return lists


def func_loss(hf, tt_pair_index, tt_sim, cache=None):
    """`cache`: any object that lives as long as the pairs do (the batch): the grouped pair lists are built once and kept on it."""
    return FuncLossFn.apply(hf, tt_pair_index, tt_sim, _pair_lists_for(hf, tt_pair_index, cache))


def func_loss_passthrough(hf, tt_pair_index, tt_sim, cache=None):
    """(loss, hf'): hf' is hf; feed hf' to the other consumer of hf (the readout) and the two gradients meet inside the
    function-loss backward kernel instead of in an N x H add."""
    return FuncLossFn.apply(hf, tt_pair_index, tt_sim, _pair_lists_for(hf, tt_pair_index, cache), True)


# ------------------------------------------------------------------------------------------------
# readout: BatchNorm1d + ReLU + Dropout block, 32 -> 1 head with clamp
# ------------------------------------------------------------------------------------------------
class BnReluDropFn(torch.autograd.Function):
    """dropout_p(relu(batch_norm(y))) for y [N,C] (mlp.py:31-36).  training=True uses batch statistics
    and updates the running buffers in place like nn.BatchNorm1d (momentum 0.1, unbiased running var)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, training, p_drop, seed, momentum, eps):
        yd = check(y.detach().contiguous(), F32, 'y')
        N, C = yd.shape
        dev = yd.device
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        if training:
            sums = torch.zeros(2 * C, dtype=torch.float64, device=dev)
            _hip.call('mgv_colstats', N, C, ptr(yd), C, ptr(sums), *_sw(yd.device))
            mean64 = sums[:C] / N
            var64 = (sums[C:] / N - mean64 * mean64).clamp_min(0.0)
            mean, var = mean64.to(F32), var64.to(F32)
            with torch.no_grad():
                running_mean.mul_(1 - momentum).add_(momentum * mean)
                running_var.mul_(1 - momentum).add_(momentum * (var64 * (N / max(N - 1, 1))).to(F32))
        else:
            mean, var = running_mean.detach().clone(), running_var.detach().clone()
        invstd = torch.rsqrt(var + eps).contiguous()
        mean = mean.contiguous()
        p = float(p_drop) if training else 0.0
        a = torch.empty_like(yd)
        _hip.call('mgv_bn_act_fwd', N, C, ptr(yd), ptr(mean), ptr(invstd), ptr(g), ptr(b), p, int(seed), ptr(a))
        ctx.save_for_backward(yd, mean, invstd, g, b)
        ctx.cfg = (bool(training), p, int(seed))
        return a

    @staticmethod
    def backward(ctx, ga):
        yd, mean, invstd, g, b = ctx.saved_tensors
        training, p, seed = ctx.cfg
        N, C = yd.shape
        ga = check(ga.contiguous(), F32, 'ga')
        dz = torch.empty_like(yd)
        sums = torch.zeros(2 * C, dtype=torch.float64, device=yd.device)
        _hip.call('mgv_bn_act_bwd', N, C, ptr(yd), ptr(mean), ptr(invstd), ptr(g), ptr(b), p, seed, ptr(ga), ptr(dz), ptr(sums), *_sw(yd.device))
        dy = torch.empty_like(yd)
        _hip.call('mgv_bn_bwd_apply', N, C, ptr(yd), ptr(mean), ptr(invstd), ptr(g), ptr(dz), ptr(sums), int(training), ptr(dy))
        return dy, sums[C:].to(F32), sums[:C].to(F32), None, None, None, None, None, None, None


class HeadFn(torch.autograd.Function):
    """clamp(a w^T + b, 0, 1): last Linear of the readout MLP + torch.clamp (dg_ae_model_aig.py:105)."""

    @staticmethod
    def forward(ctx, a, w, b, clamp01):
        ad = check(a.detach().contiguous(), F32, 'a')
        wd, bd = w.detach().contiguous().reshape(-1), b.detach().contiguous().reshape(-1)
        N, C = ad.shape
        prob = torch.empty(N, 1, dtype=F32, device=ad.device)
        _hip.call('mgv_readout_head_fwd', N, C, ptr(ad), ptr(wd), ptr(bd), int(bool(clamp01)), ptr(prob))
        ctx.save_for_backward(ad, wd, bd)
        ctx.wshape, ctx.clamp01 = w.shape, int(bool(clamp01))
        return prob

    @staticmethod
    def backward(ctx, gprob):
        ad, wd, bd = ctx.saved_tensors
        N, C = ad.shape
        gp = check(gprob.contiguous().reshape(-1), F32, 'gprob')
        da = torch.empty_like(ad)
        dw = torch.zeros_like(wd)
        db = torch.zeros_like(bd)
        _hip.call('mgv_readout_head_bwd', N, C, ptr(ad), ptr(wd), ptr(bd), ctx.clamp01, ptr(gp), ptr(da), ptr(dw), ptr(db), *_sw(ad.device))
        return da, dw.reshape(ctx.wshape), db, None


# ------------------------------------------------------------------------------------------------
# VAE sampler + KL
# ------------------------------------------------------------------------------------------------
class ReparamFn(torch.autograd.Function):
    """z = mu + exp(logstd) * eps and klsum = sum(1 + 2 logstd - mu^2 - exp(logstd)^2)
    (digvae_model.py:138-141, trainer.py:146-147)."""

    @staticmethod
    def forward(ctx, mu, logstd, eps, seed):
        mud, lsd = check(mu.detach().contiguous(), F32, 'mu'), check(logstd.detach().contiguous(), F32, 'logstd')
        n = mud.numel()
        z = torch.empty_like(mud)
        kl = torch.zeros(1, dtype=torch.float64, device=mud.device)
        if eps is None:
            eps_used = torch.empty_like(mud)
            _hip.call('mgv_reparam_fwd', n, ptr(mud), ptr(lsd), None, int(seed), ptr(eps_used), ptr(z), ptr(kl))
        else:
            eps_used = check(eps.detach().contiguous(), F32, 'eps')
            _hip.call('mgv_reparam_fwd', n, ptr(mud), ptr(lsd), ptr(eps_used), 0, None, ptr(z), ptr(kl))
        ctx.save_for_backward(mud, lsd, eps_used)
        return z, kl[0].to(F32)

    @staticmethod
    def backward(ctx, gz, gkl):
        mud, lsd, eps = ctx.saved_tensors
        dmu, dls = torch.empty_like(mud), torch.empty_like(mud)
        gzc = gz.contiguous() if gz is not None else None
        gk = gkl.detach().to(F32).reshape(1).contiguous() if gkl is not None else None
        _hip.call('mgv_reparam_bwd', mud.numel(), ptr(mud), ptr(lsd), ptr(eps), ptr(gzc), ptr(gk), 1.0, ptr(dmu), ptr(dls))
        return dmu, dls, None, None


# ------------------------------------------------------------------------------------------------
# optimiser
# ------------------------------------------------------------------------------------------------
def adam_step(param, grad, exp_avg, exp_avg_sq, lr, betas, eps, weight_decay, grad_scale, step):
    for n, t in (('param', param), ('grad', grad), ('exp_avg', exp_avg), ('exp_avg_sq', exp_avg_sq)):
        check(t, F32, n)
    _hip.call('mgv_adam_step', param.numel(), ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), float(lr),
              float(betas[0]), float(betas[1]), float(eps), float(weight_decay), float(grad_scale), int(step))
