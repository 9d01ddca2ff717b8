"""Colours after each half round of the structural encoder for a BASELINE config (how far the quotient stages could reach):
  python tools/colour_counts.py [config=2] [fraction=1.0]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multi-gate-vae_amd'))
import torch
import deepgate
from deepgate import synthetic as syn
from deepgate.graph_plan import GraphPlan
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
GraphPlan.QUOTIENT_FRACTION = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
dev = torch.device('cuda:0')
batch = deepgate.CircuitBatch.from_arrays(syn.make_batch(cfg), device=dev)
xcls = batch.x[:, 1].to(torch.uint8).contiguous()
p = GraphPlan(batch.edge_index, batch.x.shape[0])
st = p.quotient(xcls, 8)
N = batch.x.shape[0]
print('config %d, N = %d, threshold C * %.1f <= N: colours per half round' % (cfg, N, GraphPlan.QUOTIENT_FRACTION), [(s['C'], '%.1f %%' % (100.0 * s['C'] / N)) for s in st])
