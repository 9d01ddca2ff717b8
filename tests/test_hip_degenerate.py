"""GPU parity on ragged and degenerate batches: hand-built circuits of a few nodes, graphs without a single gate, gate types of the
model that do not occur, dangling inputs, graphs of very different sizes in one batch, gates wired in non-contiguous id order.
Every case runs one train-mode step (dropout 0) through Model / Trainer and is held to the oracle (`oracle.ref_cpu`) on the same
arrays: three losses, every parameter gradient.  The reference has no tests of its own for these; they are the
shapes its per-level Python loop handles implicitly (`dg_ae_model_aig.py:70-97`: an empty node mask skips the level/type)."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def hand_graph(ctype, n_inputs, gates, seed, n_pairs=None):
    """A circuit given as a gate list: `gates` = [(gate name, [fan-in node ids])], gate i gets node id n_inputs + i; levels are the
    ASAP levels (dag_utils.py:10-37), labels are random, negatives are drawn like the synthetic generator's."""
    from deepgate import synthetic as syn
    rng = np.random.Generator(np.random.PCG64(seed))
    ids = syn.GATE_IDS[ctype]
    n = n_inputs + len(gates)
    gate = np.zeros(n, dtype=np.int64)
    level = np.zeros(n, dtype=np.int64)
    src, dst = [], []
    for i, (name, fins) in enumerate(gates):
        v = n_inputs + i
        assert len(fins) == syn.FANIN[name] and all(f < v for f in fins) and len(set(fins)) == len(fins)
        gate[v] = ids[name]
        level[v] = 1 + max(level[f] for f in fins)
        for f in fins:
            src.append(f); dst.append(v)
    edge_index = np.array([src, dst], dtype=np.int64).reshape(2, -1)
    cand = np.arange(n_inputs, n) if len(gates) else np.arange(n)
    n_pairs = max(n // 4, 2) if n_pairs is None else n_pairs
    x = np.zeros((n, syn.NUM_GATE_TYPES), dtype=np.float32)
    x[np.arange(n), gate] = 1.0
    E = edge_index.shape[1]
    want = E + n
    # a graph this small may not HAVE E + n non-edges: take what exists
    free = n * (n - 1) - E
    neg = syn._negative_edges(rng, edge_index, n, min(want, max(free // 2, 0))) if free > 1 else np.zeros((2, 0), dtype=np.int64)
    return {
        'x': x, 'edge_index': edge_index, 'gate': gate.astype(np.float32).reshape(-1, 1),
        'forward_level': level, 'forward_index': np.arange(n, dtype=np.int64),
        'prob': rng.random((n, 1), dtype=np.float32), 'tt_pair_index': rng.choice(cand, size=(2, n_pairs)).astype(np.int64),
        'tt_sim': rng.random(n_pairs, dtype=np.float32), 'neg_edge_index': neg, 'num_nodes': n, 'n_gate': len(gates),
    }


def random_gates(ctype, n_inputs, n_gates, seed, names=None, local=False):
    """A random gate list over the model's gate names (or `names`); fan-ins uniform over all earlier nodes (`local`: over the last 8),
    so ids and levels are NOT contiguous per level the way the synthetic generator's are."""
    from deepgate import synthetic as syn
    rng = np.random.Generator(np.random.PCG64(seed))
    names = names or [g for g in syn.GATE_IDS[ctype] if g != 'INPUT']
    out = []
    for i in range(n_gates):
        v = n_inputs + i
        ok = [g for g in names if syn.FANIN[g] <= v]
        g = ok[int(rng.integers(len(ok)))]
        lo = max(0, v - 8) if local and v - 8 >= syn.FANIN[g] else 0
        out.append((g, [int(f) for f in lo + rng.choice(v - lo, size=syn.FANIN[g], replace=False)]))
    return out


def check_step(ctype, graphs, weights=(1.0, 4.0, 4.0), H=64, rounds=2, seed=3, loss_rtol=2e-4):
    dev = _dev()
    import deepgate
    from deepgate import ops, synthetic as syn
    from oracle import ref_cpu as R
    torch.manual_seed(seed)
    enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=H, s_rounds=rounds, t_rounds=rounds, layernorm=True)
    mod = {'aig': deepgate.dg_ae_model_aig, 'mig': deepgate.dg_ae_model_mig, 'xag': deepgate.dg_ae_model_xag, 'xmg': deepgate.dg_ae_model_xmg}[ctype]
    model = mod.Model(struct_encoder=enc, dim_hidden=H)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(dev).train()
    arrays = syn.collate(graphs)
    batch = deepgate.CircuitBatch.from_arrays(arrays, device=dev)
    tr = deepgate.Trainer(types.SimpleNamespace(model='DG_AE'), model, training_id='degenerate', save_dir='/tmp/mgv_test_exp', lr=1e-4,
                          rc_prob_func_weight=list(weights), device='cuda:0', batch_size=len(graphs), distributed=False)
    tr.optimizer.zero_grad()
    ls = tr.run_batch(batch)
    tr.weighted_loss(ls).backward()
    torch.cuda.synchronize()
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running_' not in k else v.clone()) for k, v in sd.items()}
    bn = {k: v.clone() for k, v in p.items() if 'running_' in k}
    ob = R.batch_from_arrays(lambda k: arrays[k])
    ols = R.run_batch(p, ctype, ob, training=True, bn_state=bn, p_drop=0.0, s_rounds=rounds, t_rounds=rounds)
    R.weighted_loss(ols, list(weights)).backward()
    atol_g = 1e-4 if ops.PRECISION == 'f32' else 1e-3
    for k in ('recon_loss', 'prob_loss', 'func_loss'):
        a, b = float(ls[k].detach()), float(ols[k].detach())
        assert abs(a - b) <= loss_rtol * max(abs(b), 1e-3), (k, a, b)
    for k, q in model.named_parameters():
        ref = p[k].grad
        if q.grad is None or ref is None:
            assert (ref is None or float(ref.abs().max()) < 1e-5) and (q.grad is None or float(q.grad.abs().max()) < 1e-5), k
            continue
        g, ref = q.grad.detach().cpu().numpy(), ref.numpy()
        assert np.isfinite(g).all(), k
        if 'attn_lin.weight' in k:
            g, ref = g[:, H:], ref[:, H:]
        scale = max(1e-6, float(np.abs(ref).max()))
        if k in ('readout_prob.fc.0.bias', 'readout_prob.fc.4.bias'):
            # a Linear bias in front of a BatchNorm has a mathematically zero gradient: rounding noise on both sides, measured
            # against its layer's weight gradient
            scale = float(p[k.replace('bias', 'weight')].grad.abs().max())
        np.testing.assert_allclose(g, ref, rtol=2e-3, atol=2 * atol_g * scale + 5e-6, err_msg='grad ' + k)
    return ls


def test_five_node_circuit():
    """3 inputs, one AND, one NOT: every tile, bucket and list of the plan is a fraction of its kernel's tile."""
    g = hand_graph('aig', 3, [('AND', [0, 1]), ('NOT', [3])], seed=1, n_pairs=3)
    check_step('aig', [g], weights=(1.0, 0.0, 4.0))


def test_single_gate_per_type_xmg():
    """One gate of each xmg type in a chain: five aggregators, one row each."""
    g = hand_graph('xmg', 3, [('MAJ', [0, 1, 2]), ('XOR', [3, 0]), ('AND', [4, 1]), ('OR', [5, 2]), ('NOT', [6])], seed=2, n_pairs=4)
    check_step('xmg', [g], weights=(1.0, 0.0, 4.0))


@pytest.mark.parametrize('ctype', ['aig', 'xmg'])
def test_graph_without_gates_inside_a_batch(ctype):
    """A graph that is only primary inputs (no edge, no level above 0) between two ordinary graphs; its nodes still enter the
    structural encoder, the readout's batch statistics and the negatives."""
    a = hand_graph(ctype, 6, random_gates(ctype, 6, 90, seed=4), seed=4)
    empty = hand_graph(ctype, 5, [], seed=5, n_pairs=2)
    b = hand_graph(ctype, 4, random_gates(ctype, 4, 60, seed=6), seed=6)
    check_step(ctype, [a, empty, b], weights=(1.0, 0.0, 4.0))


def test_gate_types_of_the_model_that_do_not_occur():
    """An xmg batch made of AND and NOT only (the MAJ / XOR / OR aggregators get no rows: their gradients are zero on both sides), and an
    aig batch without a single inverter."""
    g = hand_graph('xmg', 8, random_gates('xmg', 8, 150, seed=7, names=['AND', 'NOT']), seed=7)
    check_step('xmg', [g, hand_graph('xmg', 5, random_gates('xmg', 5, 40, seed=8, names=['AND']), seed=8)], weights=(1.0, 0.0, 4.0))
    g = hand_graph('aig', 8, random_gates('aig', 8, 100, seed=9, names=['AND']), seed=9)
    check_step('aig', [g], weights=(1.0, 0.0, 4.0))


def test_dangling_inputs_and_deep_chain():
    """40 inputs of which most drive nothing, then an inverter chain 300 deep (one node per level: 300 level launches of one row).
    Weights [1, 4, 0]: the chain's states converge, so the function loss would be 1 - cos of nearly equal rows — in fp32 its gradient
    is rounding noise in the REFERENCE's arithmetic too (the oracle in float64 against itself in float32: 100-1000x the gradient's
    size, `tools/deg_probe.py`), which says nothing about either side."""
    gates = [('AND', [0, 1])] + [('NOT', [40 + i]) for i in range(300)]
    g = hand_graph('aig', 40, gates, seed=10)
    check_step('aig', [g], weights=(1.0, 4.0, 0.0), rounds=1)


@pytest.mark.parametrize('ctype', ['aig', 'mig'])
def test_ragged_batch_with_scattered_ids(ctype):
    """Graphs of 9, 2,500 and 70 nodes in one batch, gates wired to arbitrary earlier nodes (a level's nodes are scattered over the id
    range; the plan's level buckets, not id ranges, decide the order)."""
    small = hand_graph(ctype, 4, random_gates(ctype, 4, 5, seed=11), seed=11, n_pairs=2)
    big = hand_graph(ctype, 100, random_gates(ctype, 100, 2400, seed=12, local=True), seed=12)
    mid = hand_graph(ctype, 10, random_gates(ctype, 10, 60, seed=13), seed=13)
    check_step(ctype, [small, big, mid], weights=(1.0, 0.0, 4.0))


def test_identical_regular_circuits_all_half_rounds_on_colours():
    """70 copies of one balanced AND tree with inverters (41,720 nodes: above the 16,384 the tests switch the quotient stages on from): the colours of the structural
    encoder stay few, so all eight half rounds of both encoders run on one row per colour and the per-node rows exist only behind the
    last one; losses and every gradient against the oracle, which knows nothing of colours."""
    gates, ids, nxt, lvl = [], list(range(256)), 256, 0
    while len(ids) > 1:
        new = []
        for a, b in zip(ids[0::2], ids[1::2]):
            gates.append(('AND', [a, b])); v = nxt; nxt += 1
            if lvl % 2 == 1:
                gates.append(('NOT', [v])); v = nxt; nxt += 1
            new.append(v)
        ids, lvl = new, lvl + 1
    graphs = [hand_graph('aig', 256, gates, seed=40 + i) for i in range(70)]
    import deepgate
    from deepgate.graph_plan import GraphPlan
    assert sum(g['num_nodes'] for g in graphs) >= GraphPlan.QUOTIENT_MIN_NODES
    check_step('aig', graphs, weights=(1.0, 0.0, 4.0), rounds=4)
    # (that the stages did run on colours)
    from deepgate import synthetic as syn
    arrays = syn.collate(graphs)
    batch = deepgate.CircuitBatch.from_arrays(arrays, device=_dev())
    plan = GraphPlan(batch.edge_index, batch.x.shape[0])
    assert len(plan.quotient(batch.x[:, 1].to(torch.uint8).contiguous(), 8)) == 8
