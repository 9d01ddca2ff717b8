"""Where does GraphPlan.quotient (colour refinement for the quotient stages) spend its time at config 2?  torch profiler over one warm build.
  python tools/quotient_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multi-gate-vae_amd'))
import torch
import deepgate
from deepgate import synthetic as syn
from deepgate.graph_plan import GraphPlan
dev = torch.device('cuda:0')
arrays = syn.make_batch(2)
batch = deepgate.CircuitBatch.from_arrays(arrays, device=dev)
xcls = batch.x[:, 1].to(torch.uint8).contiguous()
p = GraphPlan(batch.edge_index, batch.x.shape[0]); p.quotient(xcls, 8)
p = GraphPlan(batch.edge_index, batch.x.shape[0])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
t0 = time.perf_counter()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    p.quotient(xcls, 8)
    torch.cuda.synchronize()
print('build %.1f ms' % ((time.perf_counter() - t0) * 1e3))
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=18, max_name_column_width=60))
