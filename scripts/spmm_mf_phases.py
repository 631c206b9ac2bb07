"""Dev tool: phase times of the block-dense SpMM kernel (workgroup 0, wave 0; s_memrealtime stamps).
    GIST_EXTRA_FLAGS=-DMF_PROBE GIST_LIB_OUT=$PWD/gist_amd/libgist_hip_MFP.so python gist_amd/build.py
    GIST_LIB_PATH=$PWD/gist_amd/libgist_hip_MFP.so python scripts/spmm_mf_phases.py [D]"""
import ctypes, os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gist_amd import datasets, hip, _lib
from gist_amd.engine import SageEngine, dims_for
from gist_amd.sampler import EngineClusterIter
dev = torch.device('cuda', 0)
random.seed(0)
ds = datasets.reddit_synth(seed=0)
g = ds.g
it = EngineClusterIter('r', g, len(ds.par_li), 20, np.arange(g.number_of_nodes(), dtype=np.int64),
                       par_li=[p.copy() for p in ds.par_li], device=dev)
eng = SageEngine(dims_for(602, 64, 41, 1), True, 0.0, it.n_max, dev)
it.bind(eng, native=False)
b = next(iter(it))
L = _lib.load()
L.gist_mf_probe_read.argtypes = [ctypes.c_void_p]
names = {0: 'start', 1: 'rowptr in LDS (barrier)', 2: 'counts built (barrier)', 3: 'counts -> bf16 (2 barriers)'}
tile = ['tile start', 'X^T written', 'barrier', 'MFMAs issued', 'barrier', 'result tile written (barrier)', 'result rows read (barrier)', 'stores issued']
for d in [int(x) for x in (sys.argv[1:] or ['4096'])]:
    z = torch.randn(b.n, 2 * d, device=dev)
    for form in ('fwd', 'bwd'):
        for _ in range(5):
            if form == 'fwd':
                hip.spmm(b.rowptr, b.col, z[:, :d], z[:, d:], out_scale=b.norm, row_blocks=b.row_blocks)
            else:
                hip.spmm(b.t_rowptr, b.t_col, z[:, d:], z[:, :d], src_scale=b.norm, accumulate=True, row_blocks=b.row_blocks)
        torch.cuda.synchronize()
        buf = np.zeros(64, np.uint64)
        assert L.gist_mf_probe_read(buf.ctypes.data) == 0
        t = (buf.astype(np.int64) - int(buf[0])) / 100.0
        out = ['D=%d %s:' % (d, form)] + ['%s %.2f' % (names[i], t[i]) for i in range(1, 4)]
        i = 0
        while 8 + 8 * i + 7 < 64 and buf[8 + 8 * i] > buf[0] and (i == 0 or buf[8 + 8 * i] > buf[8 * i]):
            out.append(' | tile %d: ' % i + ', '.join('%s %.2f' % (tile[j], t[8 + 8 * i + j]) for j in range(8)))
            i += 1
        print(' '.join(out), flush=True)
