"""CPU tests of the dataset loader (deepgate/parser.py) against the reference's format (DG_VAE/deepgate/parser.py:22-129,
parser_func*.py): files written here in the MixGate layout, levels checked against a literal restatement of
utils/dag_utils.top_sort (:10-37)."""
import numpy as np
import pytest


def _literal_top_sort(edge_index, n):
    """dag_utils.top_sort restated literally: rounds of 'nodes none of whose parents are unevaluated'."""
    node_ids = np.arange(n)
    order = np.zeros(n, dtype=np.int64)
    uneval = np.ones(n, dtype=bool)
    parents, children = edge_index[0], edge_index[1]
    k = 0
    while uneval.any():
        unready = children[uneval[parents]]
        todo = uneval & ~np.isin(node_ids, unready)
        order[todo] = k
        uneval[todo] = False
        k += 1
    return order


def _random_dag(rng, n, n_in, max_fanin=3):
    src, dst = [], []
    gate = np.zeros(n, dtype=np.int64)
    for v in range(n_in, n):
        k = int(rng.integers(1, max_fanin + 1))
        ps = rng.choice(v, size=min(k, v), replace=False)
        gate[v] = {1: 2, 2: 3, 3: 1}[len(ps)]            # NOT / AND / MAJ ids of the non-AIG encoding
        src += list(ps)
        dst += [v] * len(ps)
    return np.array([src, dst], dtype=np.int64), gate


def test_forward_levels_match_the_reference_levelisation():
    from deepgate.parser import forward_levels
    rng = np.random.default_rng(0)
    for n in (1, 7, 60, 400):
        ei, _ = _random_dag(rng, n, max(1, n // 8))
        np.testing.assert_array_equal(forward_levels(ei, n), _literal_top_sort(ei, n))
    assert forward_levels(np.zeros((2, 0), dtype=np.int64), 5).tolist() == [0] * 5
    with pytest.raises(ValueError):
        forward_levels(np.array([[0, 1], [1, 0]]), 2)


@pytest.mark.parametrize('ctype', ['aig', 'xmg'])
def test_npz_parser_reads_the_mixgate_layout(tmp_path, ctype):
    import deepgate
    from deepgate import synthetic as syn
    rng = np.random.default_rng(1)
    circuits, labels = {}, {}
    for i in range(6):
        n = 40 + 5 * i
        ei, gate = _random_dag(rng, n, 6, max_fanin=2 if ctype == 'aig' else 3)
        if ctype == 'aig':
            gate = np.where(gate == 3, 1, gate)                     # AIG ids: PI 0, AND 1, NOT 2
        x = np.zeros((n, 3)); x[:, 0] = np.arange(n); x[:, 1] = gate
        pairs = rng.integers(6, n, size=(2, 9))
        lab = {'prob': rng.random(n), 'tt_pair_index': pairs if ctype == 'aig' else pairs.T, ('tt_sim' if ctype == 'aig' else 'tt_dis'): rng.random(9)}
        name = 'c%d' % i if i != 3 else 'dlatch'                    # a name the reference skips
        if ctype == 'aig':
            circuits[name] = dict(x=x, edge_index=ei, gate=gate.reshape(n, 1), **lab)
        else:
            circuits[name] = dict(x=x, edge_index=ei.T)
            labels[name] = lab
    circuits['empty'] = dict(circuits['c0'])
    if ctype == 'aig':
        circuits['empty']['tt_pair_index'] = np.zeros((2, 0), dtype=np.int64)[:0]
    else:
        labels['empty'] = dict(labels['c0'], tt_pair_index=[])
    cpath, lpath = tmp_path / 'graphs.npz', tmp_path / 'labels.npz'
    np.savez(cpath, circuits=np.array(circuits, dtype=object))
    np.savez(lpath, labels=np.array(labels, dtype=object))
    ds = deepgate.NpzParser(str(tmp_path), str(cpath), str(lpath), ctype, random_shuffle=False, trainval_split=0.8)
    train, val = ds.get_dataset()
    names = [g['name'] for g in train + val]
    assert names == ['c0', 'c1', 'c2', 'c4', 'c5'] and len(train) == 4 and len(val) == 1
    g = train[1]
    src = circuits['c1']
    n = src['x'].shape[0]
    assert g['x'].shape == (n, 6) and np.all(g['x'].sum(1) == 1) and np.all(g['x'].argmax(1) == src['x'][:, 1])
    ei = src['edge_index'] if ctype == 'aig' else src['edge_index'].T
    np.testing.assert_array_equal(g['edge_index'], ei)
    np.testing.assert_array_equal(g['forward_level'], _literal_top_sort(ei, n))
    np.testing.assert_array_equal(g['gate'].reshape(-1), src['x'][:, 1])
    assert g['tt_pair_index'].shape == (2, 9) and g['prob'].shape == (n, 1) and g['tt_sim'].shape == (9,)
    # second construction reads the cache and agrees
    again, _ = deepgate.NpzParser(str(tmp_path), str(cpath), str(lpath), ctype, random_shuffle=False, trainval_split=0.8).get_dataset()
    np.testing.assert_array_equal(again[1]['edge_index'], g['edge_index'])
    # the graphs batch like the synthetic ones (index fields offset, levels not)
    batch = syn.collate(train[:2])
    n0 = train[0]['num_nodes']
    assert batch['num_nodes'] == n0 + train[1]['num_nodes']
    assert batch['edge_index'][:, train[0]['edge_index'].shape[1]:].min() >= n0
    assert batch['forward_level'].max() == max(train[0]['forward_level'].max(), train[1]['forward_level'].max())


def test_parsed_graphs_carry_no_fixed_negatives(tmp_path):
    """A dataset batch must leave `neg_edge_index` unset so that negatives are drawn every step."""
    import deepgate
    from deepgate import synthetic as syn
    from deepgate.parser import parse_graph
    g = syn.make_graph('aig', 64, 7, 9, n_inputs=8)
    x = np.zeros((64, 2)); x[:, 1] = g['gate'].reshape(-1)
    p = parse_graph(x, g['edge_index'], g['prob'], g['tt_sim'], g['tt_pair_index'], 'aig', gate=g['gate'])
    assert 'neg_edge_index' not in p
    batch = deepgate.CircuitBatch.from_arrays(syn.collate([p, p]))
    assert getattr(batch, 'neg_edge_index', None) is None
    assert hasattr(deepgate.CircuitBatch.from_arrays(syn.collate([g, g])), 'neg_edge_index')     # the synthetic ones keep theirs


# ---- the reference's OWN parser output (tests/golden/g6_loader.npz, written by make_golden.py from
#      parser_func.parse_pyg_mlpgate / parser_func_others.parse_pyg_mlpgate incl. return_order_info)
@pytest.mark.parametrize('tag,ctype', [('aig', 'aig'), ('xmg', 'xmg')])
def test_parse_graph_equals_the_reference_parser_output(tag, ctype):
    import os
    from conftest import GOLDEN
    from deepgate.parser import parse_graph
    z = np.load(os.path.join(GOLDEN, 'g6_loader.npz'))
    x = z[tag + '_in_x']
    g = parse_graph(x, z[tag + '_in_edge_index'], z[tag + '_in_prob'], z[tag + '_in_tt_sim'], z[tag + '_in_tt_pair_index'], ctype,
                    gate=x[:, 1:2] if ctype == 'aig' else None)
    assert np.array_equal(g['x'], z[tag + '_x'])
    assert np.array_equal(g['edge_index'], z[tag + '_edge_index'])
    assert np.array_equal(g['forward_level'], z[tag + '_forward_level'])          # dag_utils.top_sort of the reference
    assert np.array_equal(g['forward_index'], z[tag + '_forward_index'])
    assert np.array_equal(g['tt_pair_index'], z[tag + '_tt_pair_index'])
    np.testing.assert_array_equal(g['tt_sim'], z[tag + '_tt_sim'].astype(np.float32))
    np.testing.assert_array_equal(g['prob'], z[tag + '_prob'].astype(np.float32))
    if tag + '_gate' in z.files:
        np.testing.assert_array_equal(g['gate'], z[tag + '_gate'])
    # the node ids of the fixture are shuffled: levels must not be monotone in the id (the test would otherwise be vacuous)
    assert (np.diff(g['forward_level']) < 0).any()


def test_backward_levels_of_the_reference_are_the_levels_of_the_flipped_graph():
    import os
    from conftest import GOLDEN
    from deepgate.parser import forward_levels
    z = np.load(os.path.join(GOLDEN, 'g6_loader.npz'))
    ei = z['aig_edge_index']
    assert np.array_equal(forward_levels(ei[::-1], z['aig_x'].shape[0]), z['aig_backward_level'])


def test_split_is_identical_on_every_rank_and_loader_shards_are_disjoint(tmp_path):
    """Two processes that build NpzParser independently (as the ranks of a distributed job do) must hold the same ordered
    train/val lists; GraphLoader then strides them by rank: disjoint shards that cover the list."""
    from deepgate.parser import NpzParser
    from deepgate.trainer import GraphLoader
    rng = np.random.default_rng(3)
    circuits = {}
    for k in range(23):
        n = 12 + k
        ei, gate = _random_dag(rng, n, 3, max_fanin=2)
        gate = np.where(gate == 3, 1, gate)                       # aig ids: AND = 1, NOT = 2
        x = np.stack([np.arange(n), gate], 1).astype(np.float64)
        circuits['c%02d' % k] = {'x': x, 'edge_index': ei, 'gate': gate.reshape(-1, 1).astype(np.float32),
                                 'prob': rng.random(n).astype(np.float32), 'tt_pair_index': rng.integers(0, n, (2, 5)),
                                 'tt_sim': rng.random(5).astype(np.float32)}
    np.savez(tmp_path / 'graphs.npz', circuits=np.array(circuits, dtype=object))
    a = NpzParser(str(tmp_path), str(tmp_path / 'graphs.npz'), '', 'aig')
    b = NpzParser(str(tmp_path), str(tmp_path / 'graphs.npz'), '', 'aig')       # second "rank": reads the published cache
    names = lambda lst: [g['name'] for g in lst]
    assert names(a.train_dataset) == names(b.train_dataset) and names(a.val_dataset) == names(b.val_dataset)
    assert not set(names(a.train_dataset)) & set(names(a.val_dataset))
    assert not [f for f in (tmp_path / 'inmemory_mgv').iterdir() if '.tmp.' in f.name]      # cache was published atomically
    world = 2
    shards = [GraphLoader(a.train_dataset, 2, shuffle=False, rank=r, world_size=world)._indices() for r in range(world)]
    n = len(a.train_dataset)
    assert len(shards[0]) == len(shards[1]) == (n + 1) // 2
    flat = shards[0] + shards[1]
    assert set(flat) == set(range(n))                         # every graph is trained on by exactly one rank...
    assert len(flat) - len(set(flat)) == (n + 1) // 2 * 2 - n  # ...except the wrap-around padding of DistributedSampler
