"""Embedding extraction with the MI355X path (the role of the reference's DG_VAE/examples/feature_extract*.py:9-34):
build the per-type Model, optionally load a checkpoint written by `Trainer.save` (or a reference `.pth`: same
state_dict keys), run `model(G) -> (hs, hf)` over a dataset and save the embeddings.

    python examples/feature_extract.py --type aig --data_dir DIR [--checkpoint exp/e/stage_3.pth] --out emb.npz
    python examples/feature_extract.py --type aig --synthetic 4 --out emb.npz
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepgate  # noqa: E402
from deepgate import synthetic  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--type', required=True, choices=['aig', 'mig', 'xmg', 'xag'])
    ap.add_argument('--data_dir', default='')
    ap.add_argument('--synthetic', type=int, default=0)
    ap.add_argument('--checkpoint', default='')
    ap.add_argument('--dim_hidden', type=int, default=64)
    ap.add_argument('--rounds', type=int, default=4)
    ap.add_argument('--batch_size', type=int, default=8)
    ap.add_argument('--out', default='embeddings.npz')
    a = ap.parse_args(argv)
    dev = torch.device('cuda:0')
    enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=a.dim_hidden, s_rounds=a.rounds, t_rounds=a.rounds,
                                                     enable_reverse=True, layernorm=True)
    model = getattr(deepgate, 'dg_ae_model_' + a.type).Model(struct_encoder=enc, dim_hidden=a.dim_hidden).to(dev)
    if a.checkpoint:
        model.load(a.checkpoint)
    model.eval()
    if a.synthetic > 0:
        graphs = [synthetic.make_graph(a.type, 1024, 30, 900 + i, n_inputs=64) for i in range(a.synthetic)]
    else:
        train, val = deepgate.NpzParser(a.data_dir, os.path.join(a.data_dir, 'graphs.npz'), os.path.join(a.data_dir, 'labels.npz'),
                                        a.type, random_shuffle=False, trainval_split=1.0).get_dataset()
        graphs = train + val
    out, t0 = {}, time.time()
    with torch.no_grad():
        for b0 in range(0, len(graphs), a.batch_size):
            chunk = graphs[b0:b0 + a.batch_size]
            batch = deepgate.CircuitBatch.from_arrays(synthetic.collate(chunk), device=dev)
            hs, hf = model(batch)
            ptr = batch.graph_ptr.tolist()
            for k, g in enumerate(chunk):
                name = g.get('name') or 'graph%d' % (b0 + k)
                out[name + '/hs'] = hs[ptr[k]:ptr[k + 1]].cpu().numpy()
                out[name + '/hf'] = hf[ptr[k]:ptr[k + 1]].cpu().numpy()
    torch.cuda.synchronize()
    np.savez(a.out, **out)
    print('[INFO] %d graphs embedded in %.2f s -> %s' % (len(graphs), time.time() - t0, a.out))


if __name__ == '__main__':
    main()
