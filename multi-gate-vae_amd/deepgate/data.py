"""Batch container with the field contract of the reference's `OrderedData` batches
(DG_VAE/deepgate/parser_func.py:10-40): attribute/item access, `num_nodes`, `.to(device)`.
The HIP graph plan (CSRs, level tiles) is cached on the batch."""
import numpy as np
import torch

from .graph_plan import GraphPlan

_FIELDS = ('x', 'edge_index', 'gate', 'forward_level', 'forward_index', 'prob', 'tt_pair_index', 'tt_sim',
           'neg_edge_index')


class CircuitBatch:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def __getitem__(self, k):
        return getattr(self, k)

    def __setitem__(self, k, v):
        setattr(self, k, v)

    def __contains__(self, k):
        return hasattr(self, k)

    @property
    def num_nodes(self):
        return int(self.x.shape[0])

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device, non_blocking=True))
            elif isinstance(v, GraphPlan):
                v.to(device)
        return self

    @staticmethod
    def from_arrays(arrays, device=None):
        """From the dict of numpy arrays `synthetic.collate` / `synthetic.make_batch` return."""
        b = CircuitBatch()
        for k in _FIELDS:
            if k in arrays:
                t = torch.from_numpy(np.ascontiguousarray(arrays[k]))
                setattr(b, k, t if device is None else t.to(device))
        if 'graph_ptr' in arrays:
            b.graph_ptr = torch.from_numpy(np.ascontiguousarray(arrays['graph_ptr']))
            b.num_graphs = int(b.graph_ptr.numel() - 1)
        return b


def plan_of(batch, gate_ids):
    """GraphPlan of a batch (built once, cached on the batch object; rebuilt if the gate set differs)."""
    key = tuple(int(g) for g in gate_ids)
    plan = getattr(batch, '_mgv_plan', None)
    dev = batch.edge_index.device
    if plan is None or plan.device != dev:
        plan = GraphPlan(batch.edge_index, batch.x.shape[0])
        plan.level_key = None
        batch._mgv_plan = plan
    if plan.level_key != key:
        if getattr(batch, 'forward_level', None) is None:
            # batches from a loader that skipped the host levelisation (NpzParser(levelise=False)): levels on the device
            batch.forward_level = plan.asap_levels()
            batch.forward_index = torch.arange(batch.x.shape[0], device=dev)
        plan.set_levels(batch.gate, batch.forward_level, list(key))
        # structural feature class = x[:, 1] as an integer (the one_hot(x[:,1]) quirk, dg_ae_model_aig.py:59)
        cls = batch.x[:, 1].to(torch.long)
        if bool(((cls < 0) | (cls > 5)).any()):
            raise ValueError('x[:, 1] must be a class index in [0, 6)')
        plan.xcls = cls.to(torch.uint8).contiguous()
        plan.level_key = key
    return plan
