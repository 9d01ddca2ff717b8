"""Where does the batch-plan construction spend its time (config 2)?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multi-gate-vae_amd'))
import torch
import deepgate
from deepgate import synthetic as syn
from deepgate.graph_plan import GraphPlan
dev = torch.device('cuda:0')
arrays = syn.make_batch(2)
batch = deepgate.CircuitBatch.from_arrays(arrays, device=dev)
def T(f, *a):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(*a); torch.cuda.synchronize(); return r, (time.perf_counter() - t) * 1e3
for rep in range(2):
    p, t1 = T(lambda: GraphPlan(batch.edge_index, batch.x.shape[0]))
    _, t2 = T(lambda: p.set_levels(batch.gate, batch.forward_level, [1, 2]))
    xcls = batch.x[:, 1].to(torch.uint8).contiguous()
    _, t3 = T(lambda: p.first_stage_classes(xcls))
    print('rep %d: CSRs %.1f ms, levels/tiles %.1f ms, class pairs %.1f ms' % (rep, t1, t2, t3))
    from deepgate import ops
    _, t4 = T(lambda: p.warm(xcls))
    _, t5 = T(lambda: ops.pair_lists(batch.tt_pair_index, batch.x.shape[0]))
    print('        warm (tagged lists, heavy rows) %.1f ms, pair lists %.1f ms' % (t4, t5))
    p._stage1 = None; p.__dict__.pop('_tagged', None); p.__dict__.pop('_heavy', None); p.__dict__.pop('_heavy_seg', None)
