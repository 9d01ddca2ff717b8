#!/usr/bin/env python3
"""Diagnostic: where does a wave of the PERSISTENT sweep backward kernel (csrc/sweep_persist_x3.hip) spend its cycles?
Runs the stamped build (csrc/libmgvae_diag.so, `make -C multi-gate-vae_amd/csrc diag`) on a config-2-shaped batch and prints
per-phase shares of the summed wave time, row waves and weight-gradient waves apart.  Never quote this build's run time."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multi-gate-vae_amd'))
import torch  # noqa: E402

import deepgate  # noqa: E402
from deepgate import ops, synthetic as syn  # noqa: E402

ROW = ['A: pull + attention rows -> planes', 'barrier after A', 'B: next lists issued + recompute mfma', 'C: gru backward',
       'D: three passes (weights, planes, 2 barriers, dgrad)', 'E: d(zbar) tile + 2 barriers', 'F: attention backward + row stores',
       'barrier after F + cursor', 'grid barrier (incl. store drain)']
WG = ['wait for the pass (barriers)', 'weight-gradient mfma', '', '', '', '', '', '', 'grid barrier + other phases']


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
    hf = ops.FuncSweepFn.apply(plan, hs, attn_u, Wvc, bvc, bih, bhh)
    wpack = ops.sweep_wpack(Wvc)
    ghf = torch.randn(N, H, device=dev)
    ghs = torch.empty(N, H, device=dev)
    dzb = torch.empty(N, 2 * H, device=dev)
    alpha, dsc = torch.empty(plan.E, device=dev), torch.empty(plan.E, device=dev)
    grads = [torch.zeros_like(t) for t in (attn_u, Wvc, bvc, bih, bhh)]
    roles_l = plan.persist_roles(lib.mgv_diag_sweep_persist_max_grid())
    roles = (ctypes.c_int32 * len(roles_l))(*roles_l)
    stp = (ctypes.c_int32 * len(plan.slot_tile_ptr))(*plan.slot_tile_ptr)
    sync = torch.zeros(lib.mgv_diag_sweep_persist_sync_bytes() // 4 + 4, dtype=torch.int32, device=dev)
    sticky = torch.zeros(4, dtype=torch.int32, device=dev)
    slab = torch.empty(lib.mgv_diag_sweep_persist_slab_floats(H, roles_l[T]), device=dev)
    stamps = torch.zeros(12 * 16, dtype=torch.int64, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.mgv_diag_set_persist_stamps(P(stamps))
    rc = lib.mgv_diag_func_sweep_bwd_persist_x3(
        H, ctypes.c_int64(N), T, plan.num_levels, P(plan.key_tile_ptr), roles, stp, P(plan.order), P(plan.order_span), P(plan.tile_start),
        P(plan.tile_count), P(plan.in_ptr), P(plan.in_src), P(plan.out_ptr), P(plan.out_dst), P(plan.out_slot), P(plan.gslot), P(hs), P(hf),
        P(attn_u), P(wpack), P(bvc), P(bih), P(bhh), P(ghf), P(ghs), P(dzb), P(alpha), P(dsc), *[P(g) for g in grads], P(slab),
        ctypes.c_int64(slab.numel()), 0, P(sync), P(sticky), st)
    torch.cuda.synchronize()
    assert rc == 0 and int(sticky[0]) == 0, (rc, int(sticky[0]))
    t = stamps.view(12, 16).double().cpu()
    grid = roles_l[T]
    tiles_per_wg = plan.num_tiles / grid
    row = t[:8]
    tot = row.sum()
    print('persistent backward, %d workgroups, %.1f tiles per workgroup: row waves, %.0f cycles per tile and wave' % (grid, tiles_per_wg, tot / 8 / plan.num_tiles))
    for k, name in enumerate(ROW):
        print('   %-56s %5.1f%%   %8.0f cycles per tile' % (name, 100 * row[:, k].sum() / tot, row[:, k].sum() / 8 / plan.num_tiles))
    wg = t[8:]
    tot = wg.sum()
    print('weight-gradient waves:')
    for k, name in enumerate(WG):
        if name:
            print('   %-56s %5.1f%%   %8.0f cycles per tile' % (name, 100 * wg[:, k].sum() / tot, wg[:, k].sum() / 4 / plan.num_tiles))


if __name__ == '__main__':
    main()
