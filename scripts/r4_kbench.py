#!/usr/bin/env python3
"""Dev tool (GPU box): HIP-event times of the round-4 kernels on the step's shapes.
  class layer: projection+CE only (dz = None), full, dW slabs, vs the four-launch pieces
  aggregation: fp32 block-dense (prepared) vs LDS gather vs bf16x3 block-dense, forward / backward, modes 0/1/2,
               on the Reddit-like batch and on the SAME batch with its cross-part edges removed (the dense floor)"""
import random
import sys
import os
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gist_amd import datasets, hip  # noqa: E402

DEV = torch.device('cuda', 0)


def timeit(fn, reps=60, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def class_bench():
    for (n, c, k) in ((2046, 41, 1024), (2046, 41, 512), (2046, 41, 2048), (1140, 47, 1024)):
        z = torch.randn(n, k, device=DEV)
        w = torch.randn(c, k, device=DEV) / k ** 0.5
        b = torch.randn(c, device=DEV)
        lab = torch.randint(0, c, (n,), device=DEV, dtype=torch.int32)
        ldc = (c + 3) // 4 * 4
        logits = torch.zeros(n, ldc, device=DEV)
        dlog = torch.zeros(n, ldc, device=DEV)
        rl = torch.zeros(n, device=DEV)
        dz = torch.zeros(n, k, device=DEV)
        part = torch.zeros(((n + 15) // 16) * c, device=DEV)
        from gist_amd import _lib
        slabs = torch.zeros(int(_lib.load().gist_class_dw_slab_bytes(n, c, k)) // 4, device=DEV)
        t_fwd = timeit(lambda: hip.class_layer(z, w, b, lab, n, logits[:, :c], dlog, rl, None, 0.2, 1, 0, part))
        t_all = timeit(lambda: hip.class_layer(z, w, b, lab, n, logits[:, :c], dlog, rl, dz, 0.2, 1, 0, part))
        t_all0 = timeit(lambda: hip.class_layer(z, w, b, lab, n, logits[:, :c], dlog, rl, dz, 0.0, 1, 0, part))
        t_dw = timeit(lambda: hip.class_dw_slabs(dlog[:, :c], z, slabs))
        loss = torch.zeros(1, device=DEV)
        t_g = timeit(lambda: hip.gemm_nt(z, w, b, logits[:, :c]))
        t_x = timeit(lambda: hip.softmax_xent(logits[:, :c], lab, None, n, rl, loss, dlog))
        t_n = timeit(lambda: hip.gemm_nn_dropout_(dlog[:, :c], w, dz, 0.2, 1, 0))
        dw = torch.zeros(c, k, device=DEV)
        t_t = timeit(lambda: hip.gemm_tn(dlog[:, :c], z, dw))
        print('class n=%d C=%d K=%d: fused A+B %.1f  A+B+C %.1f (p=0: %.1f)  dW slabs %.1f | pieces: gemm %.1f xent %.1f dz %.1f dW %.1f us'
              % (n, c, k, t_fwd, t_all, t_all0, t_dw, t_g, t_x, t_n, t_t), flush=True)


def spmm_bench():
    from gist_amd.sampler import EngineClusterIter
    random.seed(0)
    ds = datasets.reddit_synth(seed=0)
    g = ds.g
    it = EngineClusterIter(ds.name, g, len(ds.par_li), 20, np.arange(g.number_of_nodes(), dtype=np.int64),
                           par_li=[p.copy() for p in ds.par_li], device=DEV)

    class E(object):
        def z0_left(self, n):
            return torch.zeros(n, 602, device=DEV)

        def check_extract(self):
            pass
    it.engine = E()
    it.native = False
    batch = next(iter(it))
    n = batch.n
    nnz = int(batch.rowptr[n].item())
    rp, cl, trp, tcl, rb, norm = batch.rowptr, batch.col[:nnz], batch.t_rowptr, batch.t_col[:nnz], batch.row_blocks, batch.norm
    # the same batch without its cross-block edges
    rows = torch.repeat_interleave(torch.arange(n, device=DEV), (rp[1:] - rp[:-1]).long())
    blk = torch.bucketize(torch.arange(n, device=DEV), rb.long(), right=True) - 1
    keep = blk[rows] == blk[cl.long()]
    print('batch rows %d nnz %d, cross-block edges %d' % (n, nnz, int((~keep).sum().item())))
    rows_k, cl_k = rows[keep], cl[keep]
    rp_k = torch.zeros(n + 1, dtype=torch.int32, device=DEV)
    rp_k[1:] = torch.cumsum(torch.bincount(rows_k, minlength=n), 0).to(torch.int32)
    prep, prep_t = hip.spmm_prepare(rp, cl, rb), hip.spmm_prepare(trp, tcl, rb)
    prep_k = hip.spmm_prepare(rp_k, cl_k.contiguous(), rb)
    for d in (256, 512, 602, 1024, 2048, 4096):
        ld = d if d % 4 == 0 else d + 2
        x = torch.randn(n, ld, device=DEV)[:, :d]
        z = torch.zeros(n, 2 * d, device=DEV)
        res = {}
        for name, knob in (('dense32', 3), ('default', 0), ('lds2', 1)):
            hip.tuning('spmm_kernel', knob)
            try:
                res[name + ' fwd'] = timeit(lambda: hip.spmm(rp, cl, x, z[:, d:], out_scale=norm, row_blocks=rb, prepared=prep))
                res[name + ' fwd m1'] = timeit(lambda: hip.spmm_drop(rp, cl, x, z[:, d:], 1, 0.2, 1, d, 0, 2 * d, out_scale=norm, row_blocks=rb, prepared=prep))
                res[name + ' bwd'] = timeit(lambda: hip.spmm(trp, tcl, z[:, d:], z[:, :d], src_scale=norm, accumulate=True, row_blocks=rb, prepared=prep_t))
                if d % 4 == 0 and hip.spmm_drop_takes(2, d, z[:, d:], z[:, :d], True):
                    res[name + ' bwd m2'] = timeit(lambda: hip.spmm_drop(trp, tcl, z[:, d:], z[:, :d], 2, 0.2, 1, 0, d, 2 * d, src_scale=norm, accumulate=True, row_blocks=rb, prepared=prep_t))
                if name == 'dense32':
                    res['dense32 fwd, no cross-block edges'] = timeit(lambda: hip.spmm(rp_k, cl_k, x, z[:, d:], out_scale=norm, row_blocks=rb, prepared=prep_k))
                    for gsp in (2, 4):
                        hip.tuning('spmm_split', gsp)
                        res['dense32 fwd groups=%d' % gsp] = timeit(lambda: hip.spmm(rp, cl, x, z[:, d:], out_scale=norm, row_blocks=rb, prepared=prep))
                    hip.tuning('spmm_split', 0)
            except Exception as e:
                res[name] = repr(e)[:80]
            hip.tuning('spmm_kernel', 0)
        print('D=%d: ' % d + '  '.join('%s %s' % (k, ('%.1f' % v) if isinstance(v, float) else v) for k, v in res.items()), flush=True)
    t_prep = timeit(lambda: hip.spmm_prepare(rp, cl, rb), reps=20)
    print('prepare (one orientation, incl. allocation): %.1f us' % t_prep)


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if what in ('all', 'class'):
        class_bench()
    if what in ('all', 'spmm'):
        spmm_bench()
