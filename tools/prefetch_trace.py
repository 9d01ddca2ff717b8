#!/usr/bin/env python3
"""Device work of the batch prefetcher alone (collate + H2D + plan + per-batch caches), per batch: run under
`rocprofv3 --kernel-trace --stats` and divide the totals by the number of batches printed here.
  python tools/prefetch_trace.py [batches=8] [workers=4]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multi-gate-vae_amd'))
import torch  # noqa: E402

import deepgate  # noqa: E402
from deepgate import synthetic as syn  # noqa: E402
from deepgate.prefetch import BatchPrefetcher  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device('cuda:0')
    B = 64
    graphs = syn.make_graphs(2, batch=B)

    def chunks(k):
        for s_ in range(k):
            yield graphs[s_ % B:] + graphs[:s_ % B]
    for k in (2, n):                      # a warm-up pass, then the counted one
        pf = BatchPrefetcher(chunks(k), dev, gate_ids=[1, 2], workers=workers, skip=('neg_edge_index',))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = sum(1 for _ in pf)
        torch.cuda.synchronize()
        print('%d batches, %d workers: %.2f ms per batch' % (m, workers, (time.perf_counter() - t0) / m * 1e3))
        pf.close()


if __name__ == '__main__':
    main()
