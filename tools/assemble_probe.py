"""Cost of a batch's quotient stages from per-graph caches (GraphPlan.assemble_quotient) against the batch-level colour refinement
(GraphPlan.quotient), config 2: wall time, device time and launch count of one warm call each.  python tools/assemble_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multi-gate-vae_amd'))
import torch
import deepgate
from deepgate import synthetic as syn
from deepgate.graph_plan import GraphPlan
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
cfg = int(os.environ.get('CFG', '2'))
graphs = syn.make_graphs(cfg)
arrays = syn.collate(graphs)
batch = deepgate.CircuitBatch.from_arrays(arrays, device=dev)
xcls = batch.x[:, 1].to(torch.uint8).contiguous()
t0 = time.perf_counter()
parts = []
for g in graphs:
    ei = torch.from_numpy(g['edge_index']).to(dev)
    xc = torch.from_numpy(g['x'][:, 1].astype('uint8')).to(dev)
    parts.append(GraphPlan(ei, g['num_nodes']).quotient(xc, 8, force=True))
torch.cuda.synchronize()
print('per-graph stages of %d graphs: %.1f ms (once per graph); stages per graph: %s; colours of graph 0: %s' % (
    len(graphs), (time.perf_counter() - t0) * 1e3, sorted({len(p) for p in parts}), [s['C'] for s in parts[0]]))
node_off = arrays['graph_ptr'].tolist()
cd = __import__("deepgate.graph_plan", fromlist=["x"]).ColourDictionary()
t0 = time.perf_counter()
gcols = [[torch.from_numpy(v).to(dev) for v in cd.globals_of([dict(ptr=s["raw"]["rptr"].cpu().numpy(), ent=s["raw"]["ent"].cpu().numpy(), own=s["raw"]["own"].cpu().numpy(), xcls=s["xcls"].cpu().numpy()) for s in p])] for p in parts]
print("global colour ids through the dictionary: %.1f ms for %d graphs (once per graph); dictionary sizes %s" % ((time.perf_counter() - t0) * 1e3, len(parts), [len(m) for m in cd.maps]))
for name, fn in (("assemble_quotient_merged", lambda p: p.assemble_quotient_merged(parts, gcols, node_off, 8)), ("assemble_quotient", lambda p: p.assemble_quotient(parts, node_off, 8)), ('quotient (batch-level refinement)', lambda p: p.quotient(xcls, 8))):
    for rep in range(2):
        p = GraphPlan(batch.edge_index, batch.x.shape[0]); p.xcls = xcls
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            out = fn(p)
            torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
    ev = prof.key_averages()
    dev_ms = sum(e.self_device_time_total for e in ev) / 1e3
    n_launch = sum(e.count for e in ev if e.self_device_time_total > 0)
    st = out[8] if isinstance(out, dict) else out
    print('%-36s wall %.1f ms (under the profiler), device %.2f ms, %d device ops; colours per stage %s' % (name, wall, dev_ms, n_launch, [s['C'] for s in st]))
    if name.startswith('assemble'):
        print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=12, max_name_column_width=50))
