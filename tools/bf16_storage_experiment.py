#!/usr/bin/env python3
"""VERDICT r1 item 10 — "decide bf16 storage with data".  Emulates, in the oracle (CPU, fp32), a struct encoder that STORES every
half-round state (and the gradient flowing back through it) as one bf16 plane instead of fp32, and reports what that does to the
losses, to hs and to every parameter gradient on the BASELINE-shaped fixture g2_aig (H=64, 4+4 rounds, 4 x 256 nodes).
The other candidate of the verdict — states stored as pre-split hi/lo bf16 planes — has the numerics of the shipped bf16x3 path
and the same bytes as fp32: it saves only the VALU split (~2 % of a stage), not memory traffic.

Result on this fixture: recon loss moves by 7.5e-4, hs by 7e-3 of its scale, parameter gradients by up to 8 %
(median 2 %) — outside the 1e-4 / 1e-3 bars, so storage stays fp32.  Usage: python tools/bf16_storage_experiment.py"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'multi-gate-vae_amd'))
from oracle import ref_cpu as R  # noqa: E402


class RoundBoth(torch.autograd.Function):
    """bf16 round-trip of a stored state; its incoming gradient is stored in bf16 as well."""
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def run(z, bf16_states):
    p = R.params_from_npz(z)
    batch = R.batch_from_arrays(lambda k: z['in_' + k])
    oln = F.layer_norm
    if bf16_states:                          # a half round's output = LayerNorm(GRU(...)): round where it would be written
        F.layer_norm = lambda *a, **k: RoundBoth.apply(oln(*a, **k))
    try:
        bn = {k: v.clone() for k, v in p.items() if 'running_' in k}
        ls = R.run_batch(p, 'aig', batch, training=True, bn_state=bn, p_drop=0.0, s_rounds=4, t_rounds=4)
        R.weighted_loss(ls, [1.0, 4.0, 4.0]).backward()
    finally:
        F.layer_norm = oln
    return ls, p


def main():
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'g2_aig.npz'))
    l32, p32 = run(z, False)
    l16, p16 = run(z, True)
    keys = ('recon_loss', 'prob_loss', 'func_loss')
    print('losses fp32 states :', [float(l32[k].detach()) for k in keys])
    print('losses bf16 states :', [float(l16[k].detach()) for k in keys])
    print('|difference|       :', [abs(float(l32[k].detach()) - float(l16[k].detach())) for k in keys], '(bar: 1e-4)')
    print('hs: largest deviation %.2e of its scale' % (float((l32['hs'] - l16['hs']).abs().max()) / float(l32['hs'].abs().max())))
    dev = []
    for k, v in p32.items():
        if v.requires_grad and v.grad is not None and p16[k].grad is not None and float(v.grad.abs().max()) > 1e-6:
            dev.append((float((v.grad - p16[k].grad).abs().max()) / float(v.grad.abs().max()), k))
    dev.sort(reverse=True)
    print('parameter gradients: worst %.3f (%s), median %.3f of the tensor scale (bar: 1e-3)' % (dev[0][0], dev[0][1], float(np.median([a for a, _ in dev]))))


if __name__ == '__main__':
    main()
