"""Trainer with the reference's surface (DG_VAE/deepgate/trainer.py:20-277): same constructor
arguments, `set_training_args`, `run_batch` -> {'recon_loss','pred_bin','gt_bin','prob_loss','func_loss'},
`train`, `save`, `load`, `resume`, same checkpoint dict.  Differences, all deliberate (SURVEY.md
Appendix B): gradients ARE averaged across ranks (one RCCL all-reduce of the flat gradient buffer per
step), only rank 0 writes checkpoints, step metrics come from four device-side counters instead of a
host copy of every edge prediction, and the dead N x N negative mask of the edge split is never built.
"""
import inspect
import os
import sys
import time

import numpy as np
import torch
from torch import nn

from . import ops
from .data import CircuitBatch
from .optim import FlatAdam
from .prefetch import BatchPrefetcher
from .sampling import sorted_edge_keys
from .synthetic import collate as collate_arrays
from .utils.logger import Logger
from .utils.model_utils import load_model
from .utils.utils import AverageMeter


class GraphLoader:
    """Minimal stand-in for torch_geometric's DataLoader over a list of per-graph array dicts:
    drop_last batching, optional shuffling, DistributedSampler-style rank striding
    (trainer.py:178-195)."""

    def __init__(self, graphs, batch_size, shuffle, rank=0, world_size=1, seed=0):
        self.graphs, self.batch_size, self.shuffle = graphs, batch_size, shuffle
        self.rank, self.world_size, self.seed, self.epoch = rank, world_size, seed, 0

    def _indices(self):
        n = len(self.graphs)
        if self.shuffle or self.world_size > 1:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = torch.randperm(n, generator=g).tolist() if self.shuffle else list(range(n))
        else:
            idx = list(range(n))
        if self.world_size > 1:
            total = (n + self.world_size - 1) // self.world_size * self.world_size
            idx = (idx + idx[:total - n])[self.rank:total:self.world_size]
        return idx

    def __len__(self):
        return len(self._indices()) // self.batch_size

    def chunks(self):
        """The epoch's batches as lists of per-graph array dicts (what a BatchPrefetcher collates on its worker threads)."""
        idx = self._indices()
        self.epoch += 1
        for b in range(len(idx) // self.batch_size):
            yield [self.graphs[i] for i in idx[b * self.batch_size:(b + 1) * self.batch_size]]

    def __iter__(self):
        for chunk in self.chunks():
            yield CircuitBatch.from_arrays(collate_arrays(chunk))


class Trainer():
    def __init__(self, args, model, training_id='default', save_dir='./exp', lr=1e-4,
                 rc_prob_func_weight=[1.0, 4.0, 2.0], emb_dim=128, device='cpu', batch_size=32, num_workers=0,
                 distributed=True):
        super(Trainer, self).__init__()
        self.args = args
        self.emb_dim = emb_dim
        self.device = device
        self.lr = lr
        self.lr_step = -1
        self.rc_prob_func_weight = list(rc_prob_func_weight)
        os.makedirs(save_dir, exist_ok=True)
        self.log_dir = os.path.join(save_dir, training_id)
        os.makedirs(self.log_dir, exist_ok=True)
        time_str = time.strftime('%Y-%m-%d-%H-%M')
        self.log_path = os.path.join(self.log_dir, 'log-{}.txt'.format(time_str))
        self.batch_size = batch_size
        self.num_workers = num_workers
        self.distributed = distributed
        self.local_rank, self.rank, self.world_size = 0, 0, 1
        if self.distributed:
            if 'LOCAL_RANK' in os.environ:
                self.local_rank = int(os.environ['LOCAL_RANK'])
            backend = os.environ.get('MGV_DIST_BACKEND', 'nccl')      # 'nccl' is RCCL on ROCm; 'gloo' = single-GPU rehearsal
            ndev = torch.cuda.device_count()
            if self.local_rank >= ndev and backend == 'nccl':
                raise RuntimeError('rank with LOCAL_RANK=%d but only %d GPU(s) visible: one rank per GPU '
                                   '(MGV_DIST_BACKEND=gloo stacks ranks on the visible GPUs for rehearsals)' % (self.local_rank, ndev))
            dev_index = self.local_rank % max(ndev, 1)
            self.device = 'cuda:%d' % dev_index
            torch.cuda.set_device(dev_index)
            if not torch.distributed.is_initialized():
                # rendezvous from the torchrun environment
                if backend == 'nccl':
                    torch.distributed.init_process_group(backend='nccl', init_method='env://', device_id=torch.device(self.device))
                else:
                    torch.distributed.init_process_group(backend=backend, init_method='env://')
            self.world_size = torch.distributed.get_world_size()
            self.rank = torch.distributed.get_rank()
            print('Training in distributed mode. Device {}, Process {:}, total {:}.'.format(self.device, self.rank, self.world_size), file=sys.stderr)
        else:
            print('Training in single device: ', self.device, file=sys.stderr)
        self.reg_loss = ops.l1_loss                     # nn.L1Loss() of the reference, on the HIP kernel
        self.model = model.to(self.device)
        self.optimizer = FlatAdam(self.model.parameters(), lr=self.lr)
        if self.world_size > 1:
            for p in self.model.parameters():           # every rank starts from rank 0's weights
                torch.distributed.broadcast(p.data, 0)
            for b in self.model.buffers():
                torch.distributed.broadcast(b.data, 0)
        self.model_epoch = 0
        self.overlap = True          # second stream for the reconstruction branch
        self._side = None
        if self.local_rank == 0:
            self.logger = Logger(self.log_path)

    def set_training_args(self, rc_prob_func_weight=[], lr=-1, lr_step=-1, device='null'):
        if len(rc_prob_func_weight) == 3 and list(rc_prob_func_weight) != self.rc_prob_func_weight:
            print('[INFO] Update rc_prob_func_weight from {} to {}'.format(self.rc_prob_func_weight, rc_prob_func_weight))
            self.rc_prob_func_weight = list(rc_prob_func_weight)
        if lr > 0 and lr != self.lr:
            print('[INFO] Update learning rate from {} to {}'.format(self.lr, lr))
            self.lr = lr
            for param_group in self.optimizer.param_groups:
                param_group['lr'] = self.lr
        if lr_step > 0 and lr_step != self.lr_step:
            print('[INFO] Update learning rate step from {} to {}'.format(self.lr_step, lr_step))
            self.lr_step = lr_step
        if device != 'null' and device != self.device:
            print('[INFO] Update device from {} to {}'.format(self.device, device))
            self.device = device
            self.model = self.model.to(self.device)

    def save(self, path):
        torch.save({'epoch': self.model_epoch, 'state_dict': self.model.state_dict(),
                    'optimizer': self.optimizer.state_dict()}, path)

    def load(self, path):
        checkpoint = torch.load(path, map_location='cpu')
        self.optimizer.load_state_dict(checkpoint['optimizer'])
        for param_group in self.optimizer.param_groups:
            self.lr = param_group['lr']
        self.model_epoch = checkpoint['epoch']
        self.model.load(path)
        print('[INFO] Continue training from epoch {:}'.format(self.model_epoch))
        return path

    def resume(self):
        model_path = os.path.join(self.log_dir, 'model_last.pth')
        if os.path.exists(model_path):
            self.model, self.optimizer, self.model_epoch = load_model(self.model, model_path, optimizer=self.optimizer,
                                                                      local_rank=self.local_rank, device=self.device)
            return True
        return False

    def run_batch(self, batch, want_pred=True):
        """Forward + the three losses (trainer.py:131-174).  `general_train_test_split_edges` with zero
        val/test ratios only permutes the edges (preprocessing.py:41-50), to which the mean over edges is
        invariant: train_pos_edge_index is the batch's edge_index."""
        batch.train_pos_edge_index = batch.edge_index
        neg = getattr(batch, 'neg_edge_index', None)
        dev_is_cuda = next(self.model.parameters()).is_cuda
        if getattr(self, '_has_after_hs', None) is None:
            self._has_after_hs = 'after_hs' in inspect.signature(self.model.forward).parameters
        side = self._side_stream() if (self.overlap and dev_is_cuda and self._has_after_hs) else None

        def recon(hs, pass_hs=False):
            k = None
            if neg is None and getattr(batch, '_mgv_plan', None) is None:
                # no plan (hence no device sampler): sorted edge keys for the torch rejection sampler, static per batch
                k = getattr(batch, '_mgv_edge_keys', None)
                if k is None:
                    k = batch._mgv_edge_keys = sorted_edge_keys(batch.edge_index, batch.num_nodes)
            return self.model.recon_loss(hs, batch.train_pos_edge_index, neg, want_pred=want_pred, edge_keys=k,
                                         plan=getattr(batch, '_mgv_plan', None), **({'pass_hs': True} if pass_hs else {}))

        def recon_on_side(hs):
            # reconstruction branch (hs_decompose -> decoder loss) on a second HIP stream, recorded BEFORE the level sweep: it
            # runs beside the sweep forward, and its backward (replayed on this stream, enqueued after the sweep backward
            # because its nodes are older) beside the latency-bound sweep backward
            side.wait_event(self.model._hs_ready)
            hs.record_stream(side)
            with torch.cuda.stream(side):
                return recon(hs, pass_hs=True)

        if side is not None:
            hs, hf = self.model(batch, after_hs=recon_on_side)
            loss, pred_bin, gt_bin = self.model.after_hs_out
            self.model.after_hs_out = None          # the model must not keep this step's loss graph alive
            main = torch.cuda.current_stream()
            main.wait_stream(side)
            for t in (loss, pred_bin, gt_bin, self.model.last_confusion):
                if t is not None:
                    t.record_stream(main)
        else:
            hs, hf = self.model(batch)
            loss, pred_bin, gt_bin = recon(hs)
        loss_status = {'recon_loss': loss, 'pred_bin': pred_bin, 'gt_bin': gt_bin}
        if 'VAE' in getattr(self.args, 'model', '') and hasattr(self.model, 'kl_loss'):
            s_kl, t_kl = self.model.kl_loss()
            loss_status['kl_loss'] = s_kl + t_kl          # computed, not added (as in the reference)
        # hf has two consumers (trainer.py:155-163): the function loss first, the readout on the hf it passes through, so that the
        # two gradients meet inside the function-loss backward kernel
        loss_status['func_loss'], hf_r = ops.func_loss_passthrough(hf, batch['tt_pair_index'], batch['tt_sim'], cache=batch)
        prob = self.model.pred_prob(hf_r)
        loss_status['prob_loss'] = self.reg_loss(prob, batch['prob'])
        loss_status['confusion'] = self.model.last_confusion
        return loss_status

    def _encoder_half_rounds(self):
        """2 x rounds of the model's two structural encoders (what GraphPlan.quotient is asked for; a list), 8 when it cannot be told."""
        enc = getattr(self.model, getattr(self.model, 'ENCODER_ATTR', 'struct_encoder'), None)
        counts = {2 * int(getattr(getattr(enc, a, None), 'num_rounds', 4)) for a in ('source_conv', 'target_conv')}
        return sorted(counts)                # --s_rounds and --t_rounds may differ: both encoders' stage counts are warmed

    def _side_stream(self):
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream()
        return self._side

    def weighted_loss(self, loss_status):
        w = self.rc_prob_func_weight
        return w[0] * loss_status['recon_loss'] + w[1] * loss_status['prob_loss'] + w[2] * loss_status['func_loss']

    def train_step(self, batch, want_pred=False):
        """zero_grad -> run_batch -> weighted loss -> backward -> (all-reduce) Adam (trainer.py:222-234)."""
        self.optimizer.zero_grad()
        loss_status = self.run_batch(batch, want_pred=want_pred)
        loss = self.weighted_loss(loss_status)
        loss.backward()
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)      # backward work queued on the second stream
        self.optimizer.step()
        return loss_status

    # ---- per-step metrics read-back (trainer.py:236-244: three losses + the confusion counters), one step behind -------------
    def enqueue_metrics(self, loss_status):
        """Start the host copy of this step's 7 metrics (3 losses, TP/FP/TN/FN) into pinned memory and return the PREVIOUS
        step's values (None at the first call): the host never waits for the step it has just enqueued, so the next step's
        launches are already queued when the GPU finishes this one.  `flush_metrics()` returns the last step's values."""
        vals = torch.cat([torch.stack([loss_status['recon_loss'].detach(), loss_status['prob_loss'].detach(),
                                       loss_status['func_loss'].detach()]).double(), loss_status['confusion'].double()])
        if not vals.is_cuda:
            prev, self._m_last = getattr(self, '_m_last', None), vals.tolist()
            return prev
        if getattr(self, '_m_buf', None) is None:
            self._m_buf = torch.empty(2, 7, dtype=torch.float64).pin_memory()
            self._m_ev = [None, None]
            self._m_n = 0
        slot = self._m_n % 2
        prev = self._take_metrics(1 - slot)
        self._m_buf[slot].copy_(vals, non_blocking=True)
        self._m_ev[slot] = torch.cuda.Event()
        self._m_ev[slot].record()
        self._m_n += 1
        return prev

    def _take_metrics(self, slot):
        ev = self._m_ev[slot]
        if ev is None:
            return None
        ev.synchronize()
        self._m_ev[slot] = None
        return self._m_buf[slot].tolist()

    def flush_metrics(self):
        if getattr(self, '_m_buf', None) is None:
            prev, self._m_last = getattr(self, '_m_last', None), None
            return prev
        return self._take_metrics((self._m_n - 1) % 2)

    def _loader(self, dataset, shuffle):
        if isinstance(dataset, GraphLoader):
            return dataset
        return GraphLoader(dataset, self.batch_size, shuffle, self.rank, self.world_size)

    def train(self, num_epoch, train_dataset, val_dataset):
        train_loader = self._loader(train_dataset, shuffle=not self.distributed)
        val_loader = self._loader(val_dataset, shuffle=not self.distributed)
        batch_time = AverageMeter()
        stats = {k: AverageMeter() for k in ('recon', 'prob', 'func', 'acc', 'tp', 'fp', 'tn', 'fn')}
        print('[INFO] Start training, lr = {:.4f}'.format(self.optimizer.param_groups[0]['lr']))

        def account(vals):
            if vals is None:
                return
            tot = max(sum(vals[3:]), 1.0)
            tp, fp, tn, fn = (v / tot for v in vals[3:])
            for k, v in zip(('recon', 'prob', 'func', 'acc', 'tp', 'fp', 'tn', 'fn'),
                            (vals[0], vals[1], vals[2], tp + tn, tp, fp, tn, fn)):
                stats[k].update(v)

        for epoch in range(num_epoch):
            for phase in ['train', 'val']:
                loader = train_loader if phase == 'train' else val_loader
                self.model.train() if phase == 'train' else self.model.eval()
                # collate, host-to-device copy and plan build of the next batches run on worker threads / their own HIP streams
                # beside the current step (deepgate/prefetch.py); the reference does `batch.to(device)` inside the loop (trainer.py:223)
                batches = BatchPrefetcher(loader.chunks(), self.device, gate_ids=[g for _, g in getattr(self.model, 'GATES', [])] or None,
                                          workers=max(self.num_workers, 2), quotient_stages=self._encoder_half_rounds())
                try:
                    for iter_id, batch in enumerate(batches):
                        time_stamp = time.time()
                        if phase == 'train':
                            loss_status = self.train_step(batch)
                        else:
                            with torch.no_grad():
                                loss_status = self.run_batch(batch, want_pred=False)
                        # one small device->host copy per step (3 losses + 4 counters), read one step behind so that the host
                        # never waits for the step it has just enqueued
                        account(self.enqueue_metrics(loss_status))
                        batch_time.update(time.time() - time_stamp)
                    account(self.flush_metrics())
                finally:
                    batches.close()          # also when a step raises: worker threads, pinned staging and in-flight batches go
                if phase == 'train' and self.model_epoch % 10 == 0 and self.rank == 0:
                    self.save(os.path.join(self.log_dir, 'model_{:}.pth'.format(self.model_epoch)))
                    self.save(os.path.join(self.log_dir, 'model_last.pth'))
                if self.local_rank == 0:
                    line = '{}| Epoch: {:}/{:} |Recon: {:.4f} |ACC: {:.2f} |Prob: {:.4f} |Func: {:.4f}|Net: {:.2f}s\n'.format(
                        phase, epoch, num_epoch, stats['recon'].avg, stats['acc'].avg * 100, stats['prob'].avg,
                        stats['func'].avg, batch_time.avg)
                    self.logger.write(line)
                    print(line, end='')
            self.model_epoch += 1
            if self.lr_step > 0 and self.model_epoch % self.lr_step == 0:
                self.lr *= 0.1
                if self.local_rank == 0:
                    print('[INFO] Learning rate decay to {}'.format(self.lr))
                for param_group in self.optimizer.param_groups:
                    param_group['lr'] = self.lr
