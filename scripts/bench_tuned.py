"""Dev tool: bench.py with tuning hooks set first.  usage: bench_tuned.py knob=value [knob=value ...] -- <bench.py args>"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gist_amd import hip
i = sys.argv.index('--')
for kv in sys.argv[1:i]:
    k, v = kv.split('=')
    hip.tuning(k, float(v))
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[i + 1:]
runpy.run_path(sys.argv[0], run_name='__main__')
