#!/usr/bin/env python3
"""Diagnostic: where does a wave of the bf16x3 level-sweep backward kernel spend its cycles?
Runs the stamped build (csrc/libmgvae_diag.so, `make -C multi-gate-vae_amd/csrc diag`) over all levels of a
config-2-shaped batch and prints per-phase shares of the summed wave time.  Never quote this build's run
time (its fences forbid overlaps the real kernel has)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multi-gate-vae_amd'))
import torch  # noqa: E402

import deepgate  # noqa: E402
from deepgate import ops, synthetic as syn  # noqa: E402

FWD = ['tile meta + spans + small vectors', 'in-edge lists', 'row gathers + attention + zbar planes', 'barrier', 'mfma (weights from L2)',
       'gru epilogue -> LDS (+ barrier)', 'barrier + row stores']
BWD = ['tile meta + spans + small vectors', 'edge lists + per-edge scalars', 'pull + attention rows', 'barrier',
       'recompute mfma', 'gru backward', 'write dG planes (+2 barriers)', 'dgrad mfma', '(unused)',
       'd(zbar) tile (+2 barriers)', 'row 1 stores (after its attention math)', 'dG stores, lds atomics, barrier, slab store',
       'row 0: wait for its source rows + attention backward', 'row 1 loads issued + row 0 stores', 'row 1: wait + attention backward']


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device('cuda:0')
    lib = ctypes.CDLL(os.path.join(ROOT, 'multi-gate-vae_amd', 'csrc', 'libmgvae_diag.so'))
    arrays = syn.make_batch(2, batch=B)
    batch = deepgate.CircuitBatch.from_arrays(arrays, device=dev)
    plan = deepgate.data.plan_of(batch, [1, 2])
    N, H, T = plan.N, 64, 2
    torch.manual_seed(0)
    hs = torch.randn(N, H, device=dev)
    attn_u = torch.randn(T, 2 * H, device=dev) * 0.1
    Wvc = torch.randn(T, 3 * H, 2 * H, device=dev) * 0.1
    bvc, bih, bhh = (torch.randn(T, 3 * H, device=dev) * 0.1 for _ in range(3))
    hf = ops.func_sweep(plan, hs, attn_u, Wvc, bvc, bih, bhh) if hasattr(ops, 'func_sweep') else \
        ops.FuncSweepFn.apply(plan, hs, attn_u, Wvc, bvc, bih, bhh)
    wpack = ops.sweep_wpack(Wvc)
    ghf = torch.randn(N, H, device=dev)
    ghs = torch.empty(N, H, device=dev)
    dzb = torch.empty(N, 2 * H, device=dev)
    alpha, dsc = torch.empty(plan.E, device=dev), torch.empty(plan.E, device=dev)
    grads = [torch.zeros_like(t) for t in (attn_u, Wvc, bvc, bih, bhh)]
    ltpl = plan.level_tile_ptr
    widest = max(ltpl[i + 1] - ltpl[i] for i in range(1, len(ltpl) - 1))
    scratch = torch.empty(plan.n_active * 5 * H + widest * T * 11 * H + 256 * 6 * H * H, device=dev)
    stp = (ctypes.c_int32 * len(plan.slot_tile_ptr))(*plan.slot_tile_ptr)
    stamps = torch.zeros(8 * 16, dtype=torch.int64, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ltp = (ctypes.c_int32 * len(plan.level_tile_ptr))(*plan.level_tile_ptr)
    lib.mgv_diag_set_level_stamps(P(stamps))
    hf2 = torch.empty_like(hf)
    hf2.zero_()
    rc = lib.mgv_diag_func_sweep_fwd_x3_impl(
        H, ctypes.c_int64(N), T, plan.num_levels, ltp, P(plan.order), P(plan.order_rows), 32, P(plan.tile_start), P(plan.tile_count),
        P(plan.tile_slot), P(plan.in_ptr), P(plan.in_src), P(hs), P(hf2), P(attn_u), P(wpack), P(bvc), P(bih), P(bhh), st)
    torch.cuda.synchronize()
    assert rc == 0, rc
    t = stamps.view(8, 16).double().cpu()
    tot = t.sum()
    print('level forward: shares of summed wave cycles; %.0f cycles per tile per wave' % (tot / 8 / plan.num_tiles))
    for k, name in enumerate(FWD):
        print('   %-44s %5.1f%%' % (name, 100 * t[:, k].sum() / tot))
    stamps.zero_()
    rc = lib.mgv_diag_func_sweep_bwd_x3_impl(
        H, ctypes.c_int64(N), T, plan.num_levels, ltp, P(plan.order), P(plan.order_rows), 32, ctypes.c_int64(plan.n_active),
        P(plan.tile_start), P(plan.tile_count), P(plan.tile_slot), P(plan.slot_tiles), stp, P(plan.in_ptr), P(plan.in_src),
        P(plan.out_ptr), P(plan.out_dst), P(plan.out_slot), P(plan.gslot), P(hs), P(hf), P(attn_u), P(wpack), P(bvc), P(bih), P(bhh),
        P(ghf), P(ghs), P(dzb), P(alpha), P(dsc), *[P(g) for g in grads], P(scratch), ctypes.c_int64(scratch.numel()), 0, 0, None, None, None, None, None, None, None, 0, st)
    torch.cuda.synchronize()
    assert rc == 0, rc
    t = stamps.view(8, 16).double().cpu()
    tot = t.sum()
    ntiles = plan.num_tiles
    print('level backward: shares of summed wave cycles; %.0f cycles per tile per wave' % (tot / 8 / ntiles))
    for k, name in enumerate(BWD):
        print('   %-44s %5.1f%%   (per wave: %s)' % (name, 100 * t[:, k].sum() / tot, ' '.join('%4.1f' % (100 * v / tot * 8) for v in t[:, k])))


if __name__ == '__main__':
    main()
