"""Dataset loader with the contract of the reference's `NpzParser` (DG_VAE/deepgate/parser.py:22-129) and
`parse_pyg_mlpgate` (parser_func.py:43-69, parser_func_others.py:43-78): reads the MixGate `graphs.npz`
(`circuits` dict: per circuit `x` [n, >=2] with the gate id in column 1, `edge_index`, and for AIGs `gate`,
`prob`, `tt_pair_index`, `tt_sim`) and, for mig/xag/xmg, `labels.npz` (`labels` dict: `prob`, `tt_pair_index`,
`tt_dis`), and produces per-graph field dicts the batch container understands (`synthetic.collate` ->
`CircuitBatch`).  No torch_geometric: a graph is a dict of numpy arrays with the `OrderedData` fields
(`x` one-hot [n, 6], `edge_index` [2, e], `gate` [n, 1], `forward_level`, `forward_index`, `prob` [n, 1],
`tt_pair_index` [2, p], `tt_sim` [p]).

Levelisation follows `utils/dag_utils.top_sort` (:10-37): a node's level is the round in which all its parents have
been evaluated (ASAP level = longest path from a source); computed here in O(E) with a frontier instead of the
reference's O(levels x E) masks — same result, checked in tests against a literal restatement."""
import os

import numpy as np

# circuits the reference skips by name (parser.py:88)
SKIPPED = ('D_FF_0', 'register_cc', 'D_FF_1', 'Main_led_brightness_control_PWM', 'ProgramCounter', 'TenHertz', 'dlatch')
NUM_GATE_TYPES = 6


def forward_levels(edge_index, num_nodes):
    """ASAP levels of a DAG given as [2, E] (row 0 = parent, row 1 = child)."""
    n = int(num_nodes)
    level = np.zeros(n, dtype=np.int64)
    if edge_index.size == 0:
        return level
    src, dst = np.asarray(edge_index[0], dtype=np.int64), np.asarray(edge_index[1], dtype=np.int64)
    indeg = np.bincount(dst, minlength=n)
    order = np.argsort(src, kind='stable')
    out_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=n), out=out_ptr[1:])
    out_dst = dst[order]
    frontier = np.nonzero(indeg == 0)[0]
    done, cur = 0, 0
    while frontier.size:
        level[frontier] = cur
        done += frontier.size
        # children of the frontier lose one pending parent per edge
        starts, ends = out_ptr[frontier], out_ptr[frontier + 1]
        cnt = ends - starts
        if cnt.sum() == 0:
            break
        idx = np.repeat(starts - np.concatenate(([0], np.cumsum(cnt)[:-1])), cnt) + np.arange(cnt.sum())
        kids = out_dst[idx]
        np.subtract.at(indeg, kids, 1)
        kids = np.unique(kids)
        frontier = kids[indeg[kids] == 0]
        cur += 1
    if done != n:
        raise ValueError('edge_index is not a DAG (%d of %d nodes levelised)' % (done, n))
    return level


def parse_graph(x, edge_index, prob, tt_sim, tt_pair_index, circuit_type, gate=None, name=None, levelise=True):
    """One circuit -> field dict (parse_pyg_mlpgate).  AIG files store edge_index / tt_pair_index as [2, *]
    (parser_func.py:46-50); the other types store [*, 2] and are transposed (parser_func_others.py:47-60).
    levelise=False leaves `forward_level` / `forward_index` out: the batch is then levelised on the device
    (GraphPlan.asap_levels, csrc/plan_build.hip) when its plan is built."""
    x = np.asarray(x)
    n = x.shape[0]
    gate_id = x[:, 1].astype(np.int64)
    if gate_id.min(initial=0) < 0 or gate_id.max(initial=0) >= NUM_GATE_TYPES:
        raise ValueError('gate ids must lie in [0, %d)' % NUM_GATE_TYPES)
    feat = np.zeros((n, NUM_GATE_TYPES), dtype=np.float32)          # construct_node_feature: one-hot of x[:, 1]
    feat[np.arange(n), gate_id] = 1.0
    ei = np.asarray(edge_index, dtype=np.int64)
    tp = np.asarray(tt_pair_index, dtype=np.int64)
    if circuit_type != 'aig':
        ei = ei.T if ei.size else ei.reshape(2, 0)
        tp = tp.T if tp.size else tp.reshape(2, 0)
    ei = np.ascontiguousarray(ei.reshape(2, -1))
    tp = np.ascontiguousarray(tp.reshape(2, -1))
    g = np.asarray(gate, dtype=np.float32).reshape(n, 1) if gate is not None else x[:, 1:2].astype(np.float32)
    out = {
        'x': feat, 'edge_index': ei, 'gate': g, 'prob': np.asarray(prob, dtype=np.float32).reshape(n, 1),
        'tt_pair_index': tp, 'tt_sim': np.asarray(tt_sim, dtype=np.float32).reshape(-1),
        'num_nodes': n, 'name': name,          # no neg_edge_index: negatives are drawn every step (dg_ae_model_aig.py:115-119)
    }
    if levelise:
        out['forward_level'] = forward_levels(ei, n)
        out['forward_index'] = np.arange(n, dtype=np.int64)
    return out


class NpzParser:
    """`NpzParser(data_dir, circuit_path, label_path, circuit_type).get_dataset() -> (train, val)` lists of graphs
    (parser.py:22-41).  The parsed list is cached as `<data_dir>/inmemory_mgv/<type>.npz` like the reference's
    `inmemory/data.pt`."""

    def __init__(self, data_dir, circuit_path, label_path, circuit_type, random_shuffle=True, trainval_split=0.9, seed=0, levelise=True):
        """`seed` fixes the shuffle and therefore the train/val cut: every rank of a distributed job must hold the SAME
        ordered lists (GraphLoader strides them by rank), so the default is a constant, not entropy."""
        self.data_dir, self.circuit_type = data_dir, circuit_type
        graphs = self._load(data_dir, circuit_path, label_path, circuit_type)
        if not levelise:                      # levels then come from the device builder, batch by batch
            graphs = [{k: v for k, v in g.items() if k not in ('forward_level', 'forward_index')} for g in graphs]
        if random_shuffle:
            rng = np.random.default_rng(seed)
            graphs = [graphs[i] for i in rng.permutation(len(graphs))]
        cut = int(len(graphs) * trainval_split)
        self.train_dataset, self.val_dataset = graphs[:cut], graphs[cut:]

    def get_dataset(self):
        return self.train_dataset, self.val_dataset

    @staticmethod
    def _load(data_dir, circuit_path, label_path, circuit_type):
        cache = os.path.join(data_dir, 'inmemory_mgv', '%s.npz' % circuit_type)
        rank, dist = 0, None
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                rank = dist.get_rank()
            else:
                dist = None
        except ImportError:
            dist = None
        # With an initialised process group (call init_process_group BEFORE constructing the parser) rank 0 parses and publishes the
        # cache while the others wait; rank 0 always reaches the barrier (try/finally), so a parse error on it does not leave the
        # others waiting for the group timeout — they then fail on the missing cache.  Without a group every rank parses for
        # itself; the atomic publish below keeps that safe.
        if dist is not None and rank != 0:
            dist.barrier()
        try:
            return NpzParser._load_or_parse(cache, circuit_path, label_path, circuit_type)
        finally:
            if dist is not None and rank == 0:
                dist.barrier()

    @staticmethod
    def _load_or_parse(cache, circuit_path, label_path, circuit_type):
        if os.path.exists(cache):
            return list(np.load(cache, allow_pickle=True)['graphs'])
        circuits = np.load(circuit_path, allow_pickle=True)['circuits'].item()
        labels = None if circuit_type == 'aig' else np.load(label_path, allow_pickle=True)['labels'].item()
        tt_key = 'tt_sim' if circuit_type == 'aig' else 'tt_dis'
        graphs = []
        for name, c in circuits.items():
            if name in SKIPPED:
                continue
            lab = c if labels is None else labels[name]
            if len(lab['tt_pair_index']) == 0:
                print('No tt or rc pairs: ', name)
                continue
            graphs.append(parse_graph(c['x'], c['edge_index'], lab['prob'], lab[tt_key], lab['tt_pair_index'], circuit_type,
                                      gate=c['gate'] if circuit_type == 'aig' else None, name=name))
        os.makedirs(os.path.dirname(cache), exist_ok=True)
        arr = np.empty(len(graphs), dtype=object)
        arr[:] = graphs
        tmp = cache + '.tmp.%d.npz' % os.getpid()          # published atomically: readers never see a half-written file
        np.savez(tmp, graphs=arr)
        os.replace(tmp, cache)
        print('[INFO] Inmemory dataset save: ', cache)
        return graphs
