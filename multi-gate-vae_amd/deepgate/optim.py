"""Adam over ONE flat fp32 buffer: every parameter (and its .grad) is a view into a single
allocation, so a train step needs one RCCL all-reduce (gradient mean over ranks — the data-parallel
exchange BASELINE.json asks for; the reference itself never synchronises gradients)
and one HIP kernel launch for the update.  state_dict()/load_state_dict() speak torch.optim.Adam's
format, so `{'epoch','state_dict','optimizer'}` checkpoints (trainer.py:105-111) interchange."""
import torch
import torch.distributed as dist

from . import ops


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, sync_grads=True):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise NotImplementedError('FlatAdam keeps one parameter group')
        self.sync_grads = sync_grads
        self._flat = None
        self._step = 0
        self._pending_state = None

    # ---- flat storage -----------------------------------------------------------------------------
    def _params(self):
        return [p for p in self.param_groups[0]['params'] if p.requires_grad]

    def _is_flat(self):
        if self._flat is None:
            return False
        base = self._flat['param'].data_ptr()
        for p, off in zip(self._params(), self._flat['offsets']):
            if p.data_ptr() != base + 4 * off:
                return False
        return True

    def _flatten(self):
        ps = self._params()
        dev = ps[0].device
        if any(p.dtype != torch.float32 or p.device != dev for p in ps):
            raise ValueError('FlatAdam needs fp32 parameters on one device')
        offsets, n = [], 0
        for p in ps:
            offsets.append(n)
            n += (p.numel() + 3) // 4 * 4          # keep every view 16-byte aligned
        old = self._flat
        flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        m = torch.zeros(n, dtype=torch.float32, device=dev)
        v = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, off in zip(ps, offsets):
            k = p.numel()
            flat_p[off:off + k].copy_(p.data.reshape(-1))
            if p.grad is not None:
                flat_g[off:off + k].copy_(p.grad.reshape(-1))
            p.data = flat_p[off:off + k].view(p.shape)
            p.grad = flat_g[off:off + k].view(p.shape)
        if old is not None and old['param'].numel() == n:
            m.copy_(old['m'].to(dev))
            v.copy_(old['v'].to(dev))
        self._flat = {'param': flat_p, 'grad': flat_g, 'm': m, 'v': v, 'offsets': offsets, 'n': n}
        if self._pending_state is not None:
            self._apply_state(self._pending_state)
            self._pending_state = None

    def flat_buffers(self):
        if not self._is_flat():
            self._flatten()
        return self._flat

    def zero_grad(self, set_to_none=True):
        """Gradients are dropped, not zeroed: autograd then ASSIGNS every parameter's gradient instead of adding it to a zero-filled
        view (one small add kernel per parameter and step, 75 at the BASELINE models); `reduce_gradients` gathers them into the
        flat buffer with one multi-tensor copy."""
        if not set_to_none:
            if not self._is_flat():
                self._flatten()
            self._flat['grad'].zero_()
            for p, off in zip(self._params(), self._flat['offsets']):
                p.grad = self._flat['grad'][off:off + p.numel()].view(p.shape)
            return
        for p in self.param_groups[0]['params']:
            p.grad = None

    def _gather_grads(self):
        """The parameters' gradients -> their slots of the flat buffer (parameters without a gradient: zeros)."""
        f = self._flat
        src, dst, dead = [], [], []
        for p, off in zip(self._params(), f['offsets']):
            view = f['grad'][off:off + p.numel()].view(p.shape)
            if p.grad is None:
                dead.append(view)
            elif p.grad.data_ptr() != view.data_ptr():
                src.append(p.grad)
                dst.append(view)
        if dst:
            torch._foreach_copy_(dst, src)
        if dead:
            torch._foreach_zero_(dead)
        # from here on p.grad IS the flat view (no kernel): after the all-reduce every reader of p.grad (clipping, logging, tests)
        # sees the exchanged gradient, not this rank's local one
        for p, off in zip(self._params(), f['offsets']):
            p.grad = f['grad'][off:off + p.numel()].view(p.shape)

    @torch.no_grad()
    def reduce_gradients(self):
        """One all-reduce (sum) of the flat gradient buffer over the default process group (RCCL on the
        GPUs, gloo in the CPU tests); returns the factor that turns the sum into the mean."""
        if not self._is_flat():
            self._flatten()
        self._gather_grads()
        if self.sync_grads and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self._flat['grad'], op=dist.ReduceOp.SUM)
            return 1.0 / dist.get_world_size()
        return 1.0

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        scale = self.reduce_gradients()
        f = self._flat
        g = self.param_groups[0]
        self._step += 1
        ops.adam_step(f['param'], f['grad'], f['m'], f['v'], g['lr'], g['betas'], g['eps'], g['weight_decay'], scale, self._step)
        return loss

    # ---- torch.optim.Adam-compatible checkpoint format ----------------------------------------------
    def _indexed(self):
        """(index in the parameter group, parameter, flat offset) of every trained parameter.  torch.optim.Adam numbers
        its state by position in the group, frozen parameters included: the checkpoint format follows that."""
        offs = iter(self._flat['offsets'])
        return [(i, p, next(offs)) for i, p in enumerate(self.param_groups[0]['params']) if p.requires_grad]

    def state_dict(self):
        state = {}
        if self._flat is not None and self._step > 0:
            for i, p, off in self._indexed():
                k = p.numel()
                state[i] = {'step': torch.tensor(float(self._step)),
                            'exp_avg': self._flat['m'][off:off + k].view(p.shape).clone(),
                            'exp_avg_sq': self._flat['v'][off:off + k].view(p.shape).clone()}
        group = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        group['params'] = list(range(len(self.param_groups[0]['params'])))
        return {'state': state, 'param_groups': [group]}

    def _apply_state(self, state):
        for i, p, off in self._indexed():
            st = state.get(i, state.get(str(i)))
            if st is None:
                continue
            k = p.numel()
            self._flat['m'][off:off + k].copy_(st['exp_avg'].reshape(-1))
            self._flat['v'][off:off + k].copy_(st['exp_avg_sq'].reshape(-1))
            self._step = max(self._step, int(float(st['step'])))

    def load_state_dict(self, sd):
        g = sd['param_groups'][0]
        for k in ('lr', 'betas', 'eps', 'weight_decay'):
            if k in g:
                self.param_groups[0][k] = tuple(g[k]) if k == 'betas' else g[k]
        if self._flat is not None and self._is_flat():
            self._apply_state(sd['state'])
        else:
            # applied at the first flatten; copied now: torch.optim state_dicts alias the live optimiser's tensors
            self._pending_state = {k: {kk: (vv.detach().clone() if torch.is_tensor(vv) else vv) for kk, vv in st.items()}
                                   for k, st in sd['state'].items()}
