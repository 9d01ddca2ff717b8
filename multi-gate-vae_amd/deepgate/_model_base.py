"""Shared implementation of the four reference `Model` classes
(DG_VAE/deepgate/dg_ae_model_{aig,mig,xag,xmg}.py): structural encoding -> hs_linear -> levelised
per-gate-type attention + GRU sweep -> (hs, hf); readout, reconstruction loss, checkpoint loading.
Sub-module names and construction order follow the reference so state_dicts and seeded
initialisation line up."""
import os

import torch
from torch import nn

from . import ops
from .arch.mlp import MLP
from .arch.tfmlp import TFMlpAggr
from .data import plan_of
from .digae_layer import DirectedInnerProductDecoder
from .sampling import NegativeEdges, negative_sampling, negative_sampling_device

EPS = 1e-15
DEVICE_SAMPLER = True             # fused device sampler whenever the batch has a plan (the torch rejection sampler serves callers without one)
MAX_LOGSTD = 10


class FunctionalModel(nn.Module):
    ENCODER_ATTR = 'struct_encoder'
    GATES = ()            # ((name, gate id), ...) in the order the reference creates aggr_*/update_* modules

    def __init__(self, struct_encoder, num_rounds=1, dim_hidden=128, enable_encode=True, enable_reverse=True):
        super().__init__()
        setattr(self, self.ENCODER_ATTR, struct_encoder)
        self.decoder = DirectedInnerProductDecoder()
        self.hs_linear = nn.Linear(dim_hidden * 2, dim_hidden)
        self.hs_decompose = nn.Linear(dim_hidden, dim_hidden * 2)
        self.num_rounds = num_rounds
        self.enable_encode = enable_encode
        self.enable_reverse = enable_reverse
        self.dim_hidden = dim_hidden
        self.dim_mlp = 32
        for name, _ in self.GATES:
            setattr(self, 'aggr_%s_func' % name, TFMlpAggr(dim_hidden * 2, dim_hidden))
        for name, _ in self.GATES:
            setattr(self, 'update_%s_func' % name, nn.GRU(dim_hidden, dim_hidden))
        self.readout_prob = MLP(dim_hidden, self.dim_mlp, 1, num_layer=3, p_drop=0.2, norm_layer='batchnorm',
                                act_layer='relu')
        self.last_confusion = None

    # ---- forward ---------------------------------------------------------------------------------
    def _sweep_params(self):
        parts = [getattr(self, 'aggr_%s_func' % n).composed(getattr(self, 'update_%s_func' % n)) for n, _ in self.GATES]
        return [torch.stack([p[i] for p in parts]) for i in range(5)]

    def forward(self, G, after_hs=None):
        """`after_hs(hs)` (optional): called once hs exists and BEFORE the level sweep is recorded; its result is left in
        `self.after_hs_out`.  Trainer.run_batch starts the reconstruction branch there: its autograd nodes are then older than
        the sweep's, so the backward engine enqueues the sweep backward first and the branch's backward (on its own stream)
        runs beside it instead of beside the bandwidth-bound readout backward."""
        if self.num_rounds < 1:
            raise ValueError('num_rounds must be >= 1')
        plan = plan_of(G, [gid for _, gid in self.GATES])
        dev = self.hs_linear.weight.device
        rows = torch.eye(6, dtype=torch.float32, device=dev)          # one_hot(x[:,1], 6) rows
        enc = getattr(self, self.ENCODER_ATTR)
        s, t = enc(None, None, G.edge_index, plan=plan, classes=(rows, plan.xcls))
        hs = ops.linear(s, self.hs_linear.weight, self.hs_linear.bias, x2=t)
        # the reconstruction branch only needs hs: Trainer.run_batch may start it on a second stream as soon
        # as this event has fired, next to the (latency-bound, GPU-underfilling) level sweep
        self._hs_ready = torch.cuda.Event()
        self._hs_ready.record()
        self._hs_pass = None
        self.after_hs_out = after_hs(hs) if after_hs is not None else None
        # if the reconstruction branch ran in after_hs, hs came back through its hs_decompose node (same tensor): the sweep's
        # gradient then reaches hs inside that Linear's input-gradient kernel instead of through a separate N x H add
        hs_in = self._hs_pass if self._hs_pass is not None else hs
        self._hs_pass = None
        hf = ops.FuncSweepFn.apply(plan, hs_in, *self._sweep_params())
        # further rounds (dg_ae_model_aig.py:70; the reference default and train.py use 1): every gate is updated again, its GRU
        # starting from the gate's previous state, on the same level kernels (ops.FuncSweepRoundFn)
        for _ in range(self.num_rounds - 1):
            au, Wvc, bvc, bih, _ = self._sweep_params()
            hf = ops.FuncSweepRoundFn.apply(plan, hs_in, hf, self._round_gh(plan, hf), au, Wvc, bvc, bih)
        return hs, hf

    def _round_gh(self, plan, hf):
        """gh[N, 3H] = W_hh h_prev + b_hh with each updated node's OWN aggregator weights (nn.GRU gate order r, z, n): at H = 64 in bf16x3
        mode one grouped Linear launch over the sweep's tiles; otherwise on the plain linear kernels: the rows of every gate type are
        gathered, multiplied per gate block and put back (index moves only)."""
        H = self.dim_hidden
        if hf.is_cuda and H == 64 and ops.use_x3(H) and ops.GROUPED_ROUND:
            # one grouped Linear over the sweep's (level, slot) tiles, each tile with its slot's weights (ops.RoundGhFn)
            grus = [getattr(self, 'update_%s_func' % name) for name, _ in self.GATES]
            return ops.RoundGhFn.apply(plan, hf, torch.stack([g.weight_hh_l0 for g in grus]), torch.stack([g.bias_hh_l0 for g in grus]))
        gh = torch.zeros(hf.shape[0], 3 * H, dtype=hf.dtype, device=hf.device)
        for (name, _), idx in zip(self.GATES, plan.slot_nodes()):
            if idx.numel() == 0:
                continue
            gru = getattr(self, 'update_%s_func' % name)
            rows = hf.index_select(0, idx)
            parts = [ops.linear(rows, gru.weight_hh_l0[g * H:(g + 1) * H], gru.bias_hh_l0[g * H:(g + 1) * H]) for g in range(3)]
            gh = gh.index_copy(0, idx, torch.cat(parts, dim=1))
        return gh

    def pred_prob(self, hf, seed=None):
        return self.readout_prob(hf, clamp01=True, seed=seed)

    def recon_loss(self, hs, pos_edge_index, neg_edge_index=None, want_pred=True, edge_keys=None, plan=None, pass_hs=False):
        """`plan` (optional): the batch's GraphPlan when pos_edge_index is the batch's own edge set (any
        order) — the positive half of the backward then needs no atomics.  `pass_hs`: leave hs, passed through the
        hs_decompose node, in `self._hs_pass` for the level sweep (see forward)."""
        if pass_hs and hs.requires_grad:
            st, self._hs_pass = ops.linear_passthrough(hs, self.hs_decompose.weight, self.hs_decompose.bias)
        else:
            st = ops.linear(hs, self.hs_decompose.weight, self.hs_decompose.bias)
        if plan is not None and (plan.E != pos_edge_index.shape[1] or plan.N != hs.shape[0]):
            plan = None
        neg_csr = None
        if neg_edge_index is None:
            # with the batch's plan: fused device sampler, pairs bucketed for an atomic-free backward
            neg_edge_index = negative_sampling_device(plan) if plan is not None and hs.is_cuda and hs.shape[0] >= 2 and DEVICE_SAMPLER \
                else negative_sampling(pos_edge_index, hs.shape[0], keys=edge_keys)
        if plan is not None and hs.is_cuda and torch.is_tensor(neg_edge_index) and neg_edge_index.shape[1] > 0:
            # given negatives: bucket them once (cached on the tensor's identity) so that their gradient needs no atomics either
            # kept on the batch's plan (like its pair lists), not on the model: several batches with fixed negatives each keep theirs
            cache = getattr(plan, '_neg_cache', None)
            key = (neg_edge_index.data_ptr(), neg_edge_index._version, tuple(neg_edge_index.shape), hs.shape[0])
            if cache is None or cache[0] is not neg_edge_index or cache[1] != key:      # same tensor object, not written since
                from .sampling import bucket_negatives
                cache = plan._neg_cache = (neg_edge_index, key, bucket_negatives(neg_edge_index.long(), hs.shape[0]))
            neg_edge_index = cache[2]
        if isinstance(neg_edge_index, NegativeEdges):
            neg_csr, neg_edge_index = neg_edge_index.csr, neg_edge_index.edge_index
        loss, counts, pred_bin = ops.ReconLossFn.apply(st, pos_edge_index, neg_edge_index, want_pred, plan, neg_csr)
        self.last_confusion = counts        # {TP, FP, TN, FN} on device, no host copy needed for metrics
        Ep, En = pos_edge_index.shape[1], neg_edge_index.shape[1]
        gt_bin = None
        if want_pred:
            gt_bin = torch.zeros(Ep + En, dtype=torch.int32, device=hs.device)
            gt_bin[:Ep] = 1
        return loss, pred_bin if want_pred else None, gt_bin

    # ---- checkpoints (dg_ae_model_aig.py:132-160) --------------------------------------------------
    def load(self, model_path):
        checkpoint = torch.load(model_path, map_location=lambda storage, loc: storage)
        src = checkpoint['state_dict']
        state = {}
        for k, v in src.items():
            state[k[7:] if k.startswith('module') and not k.startswith('module_list') else k] = v
        own = self.state_dict()
        for k in list(state):
            if k in own:
                if state[k].shape != own[k].shape:
                    print('Skip loading parameter {}, required shape{}, loaded shape{}.'.format(k, own[k].shape, state[k].shape))
                    state[k] = own[k]
            else:
                print('Drop parameter {}.'.format(k))
        for k in own:
            if k not in state:
                print('No param {}.'.format(k))
                state[k] = own[k]
        self.load_state_dict(state, strict=False)

    def load_pretrained(self, pretrained_model_path=''):
        if pretrained_model_path == '':
            pretrained_model_path = os.path.join(os.path.dirname(__file__), 'pretrained', 'model.pth')
        self.load(pretrained_model_path)
