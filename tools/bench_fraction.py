"""bench.py with another quotient threshold (GraphPlan.QUOTIENT_FRACTION): python tools/bench_fraction.py 2.0 [bench.py arguments]."""
import os
import runpy
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'multi-gate-vae_amd'))
from deepgate.graph_plan import GraphPlan  # noqa: E402
GraphPlan.QUOTIENT_FRACTION = float(sys.argv[1])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
