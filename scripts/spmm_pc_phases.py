"""Dev tool: slot-by-slot times of the producer / consumer block-dense SpMM (workgroup 0: consumer wave 0, producer
wave 8; s_memrealtime stamps, 100 MHz).
    GIST_EXTRA_FLAGS=-DMF_PROBE GIST_LIB_OUT=$PWD/gist_amd/libgist_hip_MFP.so python gist_amd/build.py
    GIST_LIB_PATH=$PWD/gist_amd/libgist_hip_MFP.so python scripts/spmm_pc_phases.py [D]"""
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
prep, prep_t = hip.spmm_prepare(b.rowptr, b.col, b.row_blocks), hip.spmm_prepare(b.t_rowptr, b.t_col, b.row_blocks)
for d in [int(x) for x in (sys.argv[1:] or ['4096'])]:
    z = torch.randn(b.n, 2 * d, device=dev)
    for form in ('fwd', 'bwd'):
        for _ in range(5):
            if form == 'fwd':
                hip.spmm(b.rowptr, b.col, z[:, :d], z[:, d:], out_scale=b.norm, row_blocks=b.row_blocks, prepared=prep)
            else:
                hip.spmm(b.t_rowptr, b.t_col, z[:, d:], z[:, :d], src_scale=b.norm, accumulate=True,
                         row_blocks=b.row_blocks, prepared=prep_t)
        torch.cuda.synchronize()
        buf = np.zeros(64, np.uint64)
        assert L.gist_mf_probe_read(buf.ctypes.data) == 0
        t = (buf.astype(np.int64) - int(buf[0])) / 100.0
        print('D=%d %s (us since kernel start)' % (d, form))
        for s in range(7):
            c = t[2 + 4 * s: 6 + 4 * s]
            p = t[32 + 4 * s: 36 + 4 * s]
            print('  slot %d  consumer: A %.2f  multiplied+tile %.2f  B %.2f  rows done %.2f   |  producer: A %.2f  converted %.2f  B %.2f  loads issued %.2f'
                  % (s, c[0], c[1], c[2], c[3], p[0], p[1], p[2], p[3]), flush=True)
