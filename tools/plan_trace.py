#!/usr/bin/env python3
"""The fresh-batch plan build alone (what bench.py times as plan_ms_steady): CSRs, level tiles, packed sweep rows, heavy lists, colour
refinement of the quotient stages.  Run under `rocprofv3 --kernel-trace --stats` for its kernel list; prints wall ms per build and the
host-side phase times (each phase synchronised: the sum is larger than the pipelined build).
  python tools/plan_trace.py [builds=4] [config=2]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multi-gate-vae_amd'))
import torch  # noqa: E402

import deepgate  # noqa: E402
from deepgate import synthetic as syn  # noqa: E402
from deepgate.data import plan_of  # noqa: E402


def main():
    builds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device('cuda:0')
    arrays = syn.make_batch(cfg)
    gates = {2: [1, 2]}.get(cfg) or [g for _, g in getattr(deepgate, 'dg_ae_model_' + {3: 'mig', 5: 'xmg'}[cfg]).Model.GATES]
    rounds = 4

    def sync_ms(f):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = f()
        torch.cuda.synchronize()
        return r, (time.perf_counter() - t) * 1e3

    for i in range(builds):
        b = deepgate.CircuitBatch.from_arrays(arrays, device=dev)
        _, whole = sync_ms(lambda: plan_of(b, gates).warm(plan_of(b, gates).xcls, quotient_stages=2 * rounds))
        print('build %d: %.2f ms' % (i, whole))
    # phases, each synchronised
    from deepgate.graph_plan import GraphPlan
    for i in range(2):
        b = deepgate.CircuitBatch.from_arrays(arrays, device=dev)
        p, t_csr = sync_ms(lambda: GraphPlan(b.edge_index, b.x.shape[0]))
        _, t_lv = sync_ms(lambda: p.set_levels(b.gate, b.forward_level, gates))
        _, t_rows = sync_ms(lambda: p.order_rows)
        xcls = b.x[:, 1].to(torch.uint8).contiguous()
        _, t_heavy = sync_ms(lambda: [p.heavy(False), p.heavy(True), p.heavy_segments(False), p.heavy_segments(True),
                                      p.heavy_segments(True, inactive_only=True), p.heavy_segments(True, active_by_level=True)])
        _, t_q = sync_ms(lambda: p.quotient(xcls, 2 * rounds))
        print('phases %d: CSRs %.2f ms, levels/tiles %.2f ms, packed rows %.2f ms, heavy lists %.2f ms, quotient stages %.2f ms' %
              (i, t_csr, t_lv, t_rows, t_heavy, t_q))


if __name__ == '__main__':
    main()
