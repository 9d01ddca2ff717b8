import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_PARENT = os.path.join(ROOT, 'multi-gate-vae_amd')
for p in (ROOT, PKG_PARENT):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


import pytest  # noqa: E402


@pytest.fixture(autouse=True, scope='session')
def _quotient_stages_from_16k_nodes():
    """The product runs the early half rounds of the structural encoder on colours from 131,072 nodes per batch on (below, a step is
    launch-bound and the extra small launches cost more than the rows save: GraphPlan.QUOTIENT_MIN_NODES).  The tests keep that path
    switched on from 16,384 nodes, so that the oracle comparisons at one or two BASELINE-size graphs go through it as well."""
    from deepgate.graph_plan import GraphPlan
    old = GraphPlan.QUOTIENT_MIN_NODES
    GraphPlan.QUOTIENT_MIN_NODES = 16384
    yield
    GraphPlan.QUOTIENT_MIN_NODES = old
