"""Write a toy dataset in the MixGate npz layout (graphs.npz [+ labels.npz]) from the synthetic generator, to exercise
`train.py --data_dir`.  usage: make_toy_npz.py DIR TYPE N_GRAPHS [NODES]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multi-gate-vae_amd'))
import numpy as np
from deepgate import synthetic as syn
out, ctype, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
nodes = int(sys.argv[4]) if len(sys.argv) > 4 else 512
os.makedirs(out, exist_ok=True)
circuits, labels = {}, {}
for i in range(n):
    g = syn.make_graph(ctype, nodes, 16, 500 + i, n_inputs=nodes // 16)
    x = np.zeros((nodes, 3)); x[:, 0] = np.arange(nodes); x[:, 1] = g['gate'].reshape(-1)
    lab = {'prob': g['prob'].reshape(-1), 'tt_pair_index': g['tt_pair_index'] if ctype == 'aig' else g['tt_pair_index'].T,
           ('tt_sim' if ctype == 'aig' else 'tt_dis'): g['tt_sim']}
    if ctype == 'aig':
        circuits['toy%d' % i] = dict(x=x, edge_index=g['edge_index'], gate=g['gate'], **lab)
    else:
        circuits['toy%d' % i] = dict(x=x, edge_index=g['edge_index'].T)
        labels['toy%d' % i] = lab
np.savez(os.path.join(out, 'graphs.npz'), circuits=np.array(circuits, dtype=object))
np.savez(os.path.join(out, 'labels.npz'), labels=np.array(labels, dtype=object))
print('wrote', out)
