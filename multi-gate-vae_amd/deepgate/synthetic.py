"""Deterministic synthetic levelised circuit DAGs.

Stand-alone on purpose (numpy only, no package-relative imports): `tests/golden/make_golden.py`
loads this file by path next to the *reference* `deepgate` package, whose name our package shares.

The field contract is the one the reference's parsers produce (`DG_VAE/deepgate/parser_func.py:10-69`,
`parser_func_others.py:43-78`, `parser.py:100-119`): `x` one-hot of the gate id [N,6], `edge_index`
[2,E] (row 0 = source, row 1 = destination), `gate` [N,1], `forward_level` [N] (ASAP level, NOT offset
when graphs are batched), `forward_index` [N], `prob` [N,1], `tt_pair_index` [2,P], `tt_sim` [P].
"""
import numpy as np

# gate ids: aig uses PI0/AND1/NOT2 (dg_ae_model_aig.py:67-68); the other three circuit types use
# INPUT0/MAJ1/NOT2/AND3/OR4/XOR5 (README.md:34, dg_ae_model_xmg.py:86-90).
GATE_IDS = {
    'aig': {'INPUT': 0, 'AND': 1, 'NOT': 2},
    'mig': {'INPUT': 0, 'MAJ': 1, 'NOT': 2, 'AND': 3, 'OR': 4},
    'xag': {'INPUT': 0, 'NOT': 2, 'AND': 3, 'XOR': 5},
    'xmg': {'INPUT': 0, 'MAJ': 1, 'NOT': 2, 'AND': 3, 'OR': 4, 'XOR': 5},
}
FANIN = {'AND': 2, 'OR': 2, 'XOR': 2, 'NOT': 1, 'MAJ': 3}
DEFAULT_CYCLE = {
    'aig': ['AND', 'AND', 'NOT'],
    'mig': ['MAJ', 'MAJ', 'NOT'],
    'xag': ['AND', 'XOR', 'NOT'],
    'xmg': ['MAJ', 'XOR', 'AND', 'OR', 'NOT', 'MAJ'],
}
NUM_GATE_TYPES = 6


def make_graph(ctype, n_nodes, n_levels, seed, cycle=None, n_inputs=None):
    """One levelised DAG.  Nodes [0,P) are INPUTs (level 0); the rest are split evenly over
    `n_levels` levels with contiguous ids, so ASAP level == construction level.  Node k of a level
    takes gate `cycle[k % len(cycle)]`; its first fan-in is uniform over the previous level (pins the
    level), the others uniform over all earlier nodes, all distinct."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cycle = cycle or DEFAULT_CYCLE[ctype]
    ids = GATE_IDS[ctype]
    P = n_inputs if n_inputs is not None else max(n_nodes // 16, 1)
    per = (n_nodes - P) // n_levels
    assert per >= 1 and P + per * n_levels == n_nodes, 'n_nodes - n_inputs must divide by n_levels'
    gate = np.zeros(n_nodes, dtype=np.int64)
    level = np.zeros(n_nodes, dtype=np.int64)
    k = np.arange(per)
    gname = [cycle[i % len(cycle)] for i in range(per)]
    gid = np.array([ids[g] for g in gname], dtype=np.int64)
    fan = np.array([FANIN[g] for g in gname], dtype=np.int64)
    src_l, dst_l = [], []
    for lv in range(1, n_levels + 1):
        lo = P + (lv - 1) * per
        prev_lo, prev_hi = (0, P) if lv == 1 else (lo - per, lo)
        node = lo + k
        gate[node] = gid
        level[node] = lv
        f0 = rng.integers(prev_lo, prev_hi, size=per)
        fins = [f0]
        for extra in (1, 2):
            need = fan > extra
            if not need.any():
                break
            f = rng.integers(0, lo, size=per)
            # distinct from the fan-ins drawn so far (resample collisions; terminates because lo >= 2
            # whenever a 2-input gate exists and lo >= 3 for MAJ)
            while True:
                clash = np.zeros(per, dtype=bool)
                for g in fins:
                    clash |= (f == g)
                clash &= need
                if not clash.any():
                    break
                f[clash] = rng.integers(0, lo, size=int(clash.sum()))
            fins.append(f)
        # edge order: node-major, fan-in slot minor (like a netlist dump)
        for i, f in enumerate(fins):
            m = fan > i
            src_l.append(np.stack([node[m], np.full(int(m.sum()), i), f[m]]))
    cat = np.concatenate(src_l, axis=1)
    order = np.lexsort((cat[1], cat[0]))
    dst = cat[0][order]
    src = cat[2][order]
    edge_index = np.stack([src, dst]).astype(np.int64)
    n_gate = n_nodes - P
    n_pairs = max(n_nodes // 4, 1)
    tt_pair_index = rng.integers(P, n_nodes, size=(2, n_pairs)).astype(np.int64)
    tt_sim = rng.random(n_pairs, dtype=np.float32)
    prob = rng.random((n_nodes, 1), dtype=np.float32)
    E = edge_index.shape[1]
    neg = _negative_edges(rng, edge_index, n_nodes, E + n_nodes)
    x = np.zeros((n_nodes, NUM_GATE_TYPES), dtype=np.float32)
    x[np.arange(n_nodes), gate] = 1.0
    return {
        'x': x, 'edge_index': edge_index, 'gate': gate.astype(np.float32).reshape(-1, 1),
        'forward_level': level, 'forward_index': np.arange(n_nodes, dtype=np.int64),
        'prob': prob, 'tt_pair_index': tt_pair_index, 'tt_sim': tt_sim, 'neg_edge_index': neg,
        'num_nodes': n_nodes, 'n_gate': n_gate,
    }


def _negative_edges(rng, edge_index, n, count):
    """`count` uniform (src,dst) pairs that are neither existing edges nor self loops."""
    key = edge_index[0] * n + edge_index[1]
    key = np.sort(key)
    out = np.empty((2, 0), dtype=np.int64)
    while out.shape[1] < count:
        m = count - out.shape[1]
        s = rng.integers(0, n, size=m + 16)
        d = rng.integers(0, n, size=m + 16)
        kk = s * n + d
        bad = s == d
        if key.size:                      # (a graph of primary inputs only has no edge to collide with)
            pos = np.searchsorted(key, kk)
            pos[pos >= key.size] = key.size - 1
            bad |= key[pos] == kk
        out = np.concatenate([out, np.stack([s[~bad], d[~bad]])], axis=1)
    return out[:, :count]


def collate(graphs):
    """Batch graphs the way the reference's `OrderedData.__inc__/__cat_dim__` does
    (`parser_func.py:28-40`): every key containing "index" is offset by the running node count and
    edge/pair indices concatenate along dim 1; levels are NOT offset."""
    off = 0
    keys = ('x', 'edge_index', 'gate', 'forward_level', 'forward_index', 'prob', 'tt_pair_index', 'tt_sim', 'neg_edge_index')
    acc = {k: [] for k in keys if all(k in g for g in graphs)}      # dataset graphs carry no fixed negatives
    graph_ptr = [0]
    for g in graphs:
        for k in acc:
            v = g[k]
            acc[k].append(v + off if 'index' in k else v)
        off += g['num_nodes']
        graph_ptr.append(off)
    out = {}
    for k, vs in acc.items():
        out[k] = np.concatenate(vs, axis=1 if k in ('edge_index', 'tt_pair_index', 'neg_edge_index') else 0)
    out['num_nodes'] = off
    out['graph_ptr'] = np.asarray(graph_ptr, dtype=np.int64)
    return out


# named BASELINE.json configurations
CONFIGS = {
    1: dict(ctype='aig', batch=4, n_nodes=1024, n_inputs=64, n_levels=30),
    2: dict(ctype='aig', batch=64, n_nodes=65536, n_inputs=4096, n_levels=120),
    3: dict(ctype='mig', batch=64, n_nodes=65536, n_inputs=4096, n_levels=120),
    4: dict(ctype='aig', batch=64, n_nodes=65536, n_inputs=4096, n_levels=120),
    5: dict(ctype='xmg', batch=16, n_nodes=262144, n_inputs=16384, n_levels=240),
}


def make_graphs(config, batch=None, first_graph=0):
    """The graphs of a synthetic batch for BASELINE config `config` (1..5) as per-graph array dicts; graph g uses seed 1000*config + g."""
    c = dict(CONFIGS[config])
    b = batch if batch is not None else c['batch']
    return [make_graph(c['ctype'], c['n_nodes'], c['n_levels'], 1000 * config + first_graph + i, n_inputs=c['n_inputs']) for i in range(b)]


def make_batch(config, batch=None, first_graph=0):
    """Synthetic batch for BASELINE config `config` (1..5), collated."""
    return collate(make_graphs(config, batch, first_graph))
