"""Batch prefetcher: what stands between the dataset and `Trainer.train_step` (trainer.py:189-195,223 of the reference: a
DataLoader whose workers collate, then `batch.to(device)` inside the loop).

Per batch, on a host worker thread: the graphs of the batch are collated straight into pinned staging buffers (one pass per
field, index fields offset on the way: `OrderedData.__inc__/__cat_dim__`, parser_func.py:28-40), copied to the device on that
worker's own HIP stream, and the batch's graph plan (CSRs, level tiles: csrc/plan_build.hip) is built on the same stream.  The
consumer receives the batch together with an event; its stream waits for the event, the host never does.  `depth` batches are
in flight, so collate + H2D + plan build of batch k+1.. run beside the train step of batch k.
On a CPU device the same code path runs without streams (host-logic tests).
"""
import collections
import concurrent.futures
import queue
import threading

import numpy as np
import torch

from .data import CircuitBatch, plan_of

_CAT1 = ('edge_index', 'tt_pair_index', 'neg_edge_index')        # concatenated along dim 1, the others along dim 0
_KEYS = ('x', 'edge_index', 'gate', 'forward_level', 'forward_index', 'prob', 'tt_pair_index', 'tt_sim', 'neg_edge_index')


class _Staging:
    """Pinned host buffers of one in-flight batch, grown on demand and reused once the batch's copy has completed."""

    def __init__(self, pin):
        self.pin, self.buf, self.event = pin, {}, None

    def view(self, key, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        b = self.buf.get(key)
        if b is None or b.numel() < n:
            b = self.buf[key] = torch.empty(max(n, 1), dtype=torch.uint8, pin_memory=self.pin)
        t = b[:n].view(getattr(torch, np.dtype(dtype).name)).reshape(shape)
        return t


def collate_into(graphs, staging, keys=None):
    """Collate per-graph array dicts into tensors backed by `staging`'s buffers (same result as synthetic.collate)."""
    keys = [k for k in (keys or _KEYS) if all(k in g for g in graphs)]
    out, off, graph_ptr = {}, 0, [0]
    views = {}
    for k in keys:
        a0 = np.asarray(graphs[0][k])
        if k in _CAT1:
            shape = (a0.shape[0], sum(np.asarray(g[k]).shape[1] for g in graphs))
        else:
            shape = (sum(np.asarray(g[k]).shape[0] for g in graphs),) + a0.shape[1:]
        t = staging.view(k, shape, a0.dtype)
        views[k] = (t, t.numpy(), 0)
        out[k] = t
    for g in graphs:
        for k in keys:
            t, v, pos = views[k]
            a = np.asarray(g[k])
            n = a.shape[1] if k in _CAT1 else a.shape[0]
            dst = v[:, pos:pos + n] if k in _CAT1 else v[pos:pos + n]
            if 'index' in k:
                np.add(a, off, out=dst)
            else:
                np.copyto(dst, a)
            views[k] = (t, v, pos + n)
        off += int(g['num_nodes'])
        graph_ptr.append(off)
    out['graph_ptr'] = torch.tensor(graph_ptr, dtype=torch.int64)
    return out, off


def _record_all(v, stream, seen):
    """record_stream on every device tensor reachable from `v` (dicts, lists, tuples, objects with a __dict__: the batch, its plan
    and the plan's caches)."""
    if torch.is_tensor(v):
        if v.is_cuda:
            v.record_stream(stream)
        return
    if id(v) in seen or isinstance(v, (str, bytes, int, float, bool, type(None), np.ndarray)):
        return
    seen.add(id(v))
    if isinstance(v, dict):
        for u in v.values():
            _record_all(u, stream, seen)
    elif isinstance(v, (list, tuple)):
        if len(v) > 4096:               # host-side index tables (lists of ints)
            return
        for u in v:
            _record_all(u, stream, seen)
    elif hasattr(v, '__dict__'):
        _record_all(v.__dict__, stream, seen)


class BatchPrefetcher:
    """Iterate device-resident `CircuitBatch` objects over `chunks` (an iterable of lists of per-graph array dicts).

    gate_ids : build the GraphPlan for this aggregator set on the worker's stream (None: the model builds it on first use)
    workers  : host threads that collate / copy / plan (numpy releases the GIL in the large copies)
    depth    : batches in flight (>= workers)
    skip     : fields left out (e.g. 'neg_edge_index' when negatives are drawn on the device)
    quotient_stages : half rounds of the structural encoder (2 x rounds) whose colour classes are prepared with the plan (GraphPlan.quotient)
    """

    def __init__(self, chunks, device, gate_ids=None, workers=2, depth=None, skip=(), quotient_stages=8):
        self.chunks = iter(chunks)
        self.device = torch.device(device)
        self.cuda = self.device.type == 'cuda'
        if self.cuda and self.device.index is None:      # 'cuda': torch.cuda.set_device / Stream need an index
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.gate_ids = list(gate_ids) if gate_ids is not None else None
        self.workers = max(int(workers), 1)
        self.depth = max(int(depth) if depth is not None else self.workers + 1, self.workers)
        self.keys = [k for k in _KEYS if k not in skip]
        self.quotient_stages = quotient_stages if isinstance(quotient_stages, (tuple, list, set)) else int(quotient_stages)
        self._free = queue.Queue()
        for _ in range(self.depth + 1):
            self._free.put(_Staging(self.cuda))
        self._tls = threading.local()
        self._pool = concurrent.futures.ThreadPoolExecutor(self.workers, thread_name_prefix='mgv-prefetch')

    # ---- worker side
    def _stream(self):
        s = getattr(self._tls, 'stream', None)
        if s is None and self.cuda:
            torch.cuda.set_device(self.device)
            s = self._tls.stream = torch.cuda.Stream(self.device)
        return s

    # ---- per-graph colour refinement, cached on the graph (a dataset's graphs come back every epoch, in other batches)
    PARTS_CACHE_BYTES = 16 << 30         # device bytes of cached per-graph stages; beyond: batches refine their colours themselves
    # How a fresh batch gets its quotient stages (the early half rounds of the structural encoder on one row per colour):
    #   False      the batch refines its colours itself (GraphPlan.quotient: 9.6 ms of device time per config-2 batch)
    #   'separate' from its graphs' cached stages, colours NOT merged across graphs (assemble_quotient: cheaper plan, but 192 / 6,000 /
    #              460,531 colours in three stages instead of 3 / 152 / 34,377 / 1,662,243 in four: the step is ~3 ms slower; measured
    #              76.7 ms per fresh-batch step against 73.8)
    #   'merged'   from its graphs' cached stages, merged through the dataset-wide ColourDictionary: exactly the batch-level refinement
    #              (3 / 152 / 34,377 / 1,662,243), but the merge needs a sort of the graphs' colours, two sorts for the segment tables per
    #              stage and one of the nodes: as much device time as refining (measured 73.5 ms per fresh-batch step against 73.8)
    # Neither cached form pays at config 2: the cost of a batch's stages is building their tables, not finding the colours.  Default: False.
    PER_GRAPH_QUOTIENT = False
    _colour_dict = None
    _parts_lock = threading.Lock()
    _parts_bytes = 0

    def _graph_parts(self, graphs, stream):
        """Every graph's own quotient stages (GraphPlan.quotient(..., force=True) of the single graph), from the cache kept on the graph
        dict or computed now on this worker's stream; None when the batch cannot use them."""
        from . import ops
        from .graph_plan import GraphPlan
        counts = self.quotient_stages if isinstance(self.quotient_stages, (tuple, list, set)) else [self.quotient_stages]
        max_st = max([int(c) for c in counts] + [0])
        if not (self.PER_GRAPH_QUOTIENT and ops.QUOTIENT and max_st > 0 and all(isinstance(g, dict) and 'edge_index' in g and 'x' in g for g in graphs)):
            return None
        if sum(int(g['num_nodes']) for g in graphs) < GraphPlan.QUOTIENT_MIN_NODES:
            return None
        key = (str(self.device), max_st, self.PER_GRAPH_QUOTIENT)
        parts, gcols = [], []
        for g in graphs:
            hit = g.get('_mgv_quot', {}).get(key) if isinstance(g.get('_mgv_quot'), dict) else None
            if hit is None:
                with BatchPrefetcher._parts_lock:
                    hit = g.get('_mgv_quot', {}).get(key) if isinstance(g.get('_mgv_quot'), dict) else None
                    if hit is None:
                        if BatchPrefetcher._parts_bytes > self.PARTS_CACHE_BYTES:
                            return None
                        n = int(g['num_nodes'])
                        ei = torch.from_numpy(np.ascontiguousarray(g['edge_index'])).to(self.device, non_blocking=True)
                        xc = torch.from_numpy(np.ascontiguousarray(np.asarray(g['x'])[:, 1]).astype(np.uint8)).to(self.device, non_blocking=True)
                        st = GraphPlan(ei, n).quotient(xc, max_st, force=True)
                        st = [dict(C=s_['C'], cid=s_['cid'], rev=s_['rev'], xcls=s_['xcls'], raw=s_['raw']) for s_ in st]
                        gcol = None
                        if self.PER_GRAPH_QUOTIENT == 'merged':
                            # global ids of the graph's colours: the representatives' signatures through the dataset-wide dictionary
                            # (host side, once per graph: the only read-back of this path)
                            if BatchPrefetcher._colour_dict is None:
                                from .graph_plan import ColourDictionary
                                BatchPrefetcher._colour_dict = ColourDictionary()
                            host = [dict(ptr=s_['raw']['rptr'].cpu().numpy(), ent=s_['raw']['ent'].cpu().numpy(),
                                         own=s_['raw']['own'].cpu().numpy(), xcls=s_['xcls'].cpu().numpy()) for s_ in st]
                            gcol = [torch.from_numpy(v).to(self.device, non_blocking=True) for v in BatchPrefetcher._colour_dict.globals_of(host)]
                        ev = torch.cuda.Event()
                        ev.record(stream)
                        nbytes = 0
                        for s_ in st:
                            for v in list(s_['raw'].values()) + [s_['cid'], s_['xcls']]:
                                for t in (v if isinstance(v, tuple) else (v,)):
                                    if torch.is_tensor(t):
                                        nbytes += t.numel() * t.element_size()
                        BatchPrefetcher._parts_bytes += nbytes
                        hit = (st, ev, gcol)
                        g.setdefault('_mgv_quot', {})[key] = hit
            stream.wait_event(hit[1])        # (computed on another worker's stream, possibly)
            parts.append(hit[0])
            gcols.append(hit[2])
        return parts, (gcols if self.PER_GRAPH_QUOTIENT == 'merged' and all(c is not None for c in gcols) else None)

    def _take_staging(self):
        st = self._free.get()
        if st.event is not None:             # its previous batch's host-to-device copy (long finished by the time the slot comes round)
            st.event.synchronize()
            st.event = None
        return st

    def _produce(self, graphs):
        st = self._take_staging()
        host, _ = collate_into(graphs, st, self.keys)
        b = CircuitBatch()
        if not self.cuda:
            for k, t in host.items():
                setattr(b, k, t.clone() if k != 'graph_ptr' else t)
            b.num_graphs = len(graphs)
            self._free.put(st)
            if self.gate_ids is not None:
                plan_of(b, self.gate_ids)
            return b, None
        stream = self._stream()
        with torch.cuda.stream(stream):
            for k, t in host.items():
                setattr(b, k, t if k == 'graph_ptr' else t.to(self.device, non_blocking=True))
            b.num_graphs = len(graphs)
            copied = torch.cuda.Event()
            copied.record(stream)
            st.event = copied
            self._free.put(st)
            if self.gate_ids is not None:
                plan = plan_of(b, self.gate_ids)        # its few host read-backs wait on THIS stream only
                got = self._graph_parts(graphs, stream)
                if got is not None:
                    # the batch's quotient stages from its graphs' cached ones instead of a colour refinement per batch
                    if got[1] is not None:
                        plan.assemble_quotient_merged(got[0], got[1], host['graph_ptr'].tolist(), self.quotient_stages)
                    else:
                        plan.assemble_quotient(got[0], host['graph_ptr'].tolist(), self.quotient_stages)
                plan.warm(plan.xcls, self.quotient_stages)     # ... and those of the caches the step would build lazily
                if getattr(b, 'tt_pair_index', None) is not None and b.tt_pair_index.shape[1] >= 2:
                    from . import ops
                    b._mgv_pair_lists = ops.pair_lists(b.tt_pair_index, b.x.shape[0])
            ready = torch.cuda.Event()
            ready.record(stream)
        return b, ready

    # ---- consumer side
    def __iter__(self):
        pending = collections.deque()
        try:
            while True:
                while len(pending) < self.depth:
                    try:
                        graphs = next(self.chunks)
                    except StopIteration:
                        break
                    pending.append(self._pool.submit(self._produce, graphs))
                if not pending:
                    return
                b, ready = pending.popleft().result()
                if ready is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ready)
                    # the tensors were allocated on the worker's stream: tell the allocator that this stream uses them too
                    _record_all(b.__dict__, cur, set())
                yield b
        finally:
            for f in pending:
                f.cancel()

    def close(self):
        self._pool.shutdown(wait=True, cancel_futures=True)
