"""Readout MLP (reference: DG_VAE/deepgate/arch/mlp.py:14-56): [Linear, BatchNorm1d, ReLU, Dropout] x
(num_layer-1) + Linear.  Same `fc` Sequential layout, so state_dict keys match (fc.0, fc.1, fc.4, ...);
forward runs the MFMA Linear kernel and the fused BN/ReLU/Dropout and head kernels."""
import torch
import torch.nn as nn

from .. import ops


class MLP(nn.Module):
    def __init__(self, dim_in=256, dim_hidden=32, dim_pred=1, num_layer=3, norm_layer=None, act_layer=None,
                 p_drop=0.5, sigmoid=False, tanh=False):
        super().__init__()
        assert num_layer >= 2, 'The number of layers shoud be larger or equal to 2.'
        if norm_layer != 'batchnorm' or act_layer != 'relu' or sigmoid or tanh or dim_pred != 1 or p_drop <= 0:
            raise NotImplementedError('the HIP readout implements Linear-BatchNorm1d-ReLU-Dropout blocks with a '
                                      'scalar output (the configuration the DG_AE models use)')
        fc = [nn.Linear(dim_in, dim_hidden), nn.BatchNorm1d(dim_hidden), nn.ReLU(inplace=True), nn.Dropout(p_drop)]
        for _ in range(num_layer - 2):
            fc += [nn.Linear(dim_hidden, dim_hidden), nn.BatchNorm1d(dim_hidden), nn.ReLU(inplace=True), nn.Dropout(p_drop)]
        fc.append(nn.Linear(dim_hidden, dim_pred))
        self.fc = nn.Sequential(*fc)
        self.num_blocks = num_layer - 1
        self._step = 0

    def forward(self, x, clamp01=False, seed=None):
        """`seed` fixes the dropout masks (tests); by default one is drawn from torch's CPU generator."""
        y = x
        for k in range(self.num_blocks):
            lin, bn, drop = self.fc[4 * k], self.fc[4 * k + 1], self.fc[4 * k + 3]
            y = ops.linear(y, lin.weight, lin.bias)
            training = self.training
            if training and drop.p > 0:
                s = int(torch.randint(0, 2 ** 62, (1,)).item()) if seed is None else int(seed) + 7919 * k
            else:
                s = 0
            if training and bn.track_running_stats:
                bn.num_batches_tracked += 1
            y = ops.BnReluDropFn.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, training,
                                       drop.p, s, bn.momentum, bn.eps)
        last = self.fc[4 * self.num_blocks]
        return ops.HeadFn.apply(y, last.weight, last.bias, clamp01)
