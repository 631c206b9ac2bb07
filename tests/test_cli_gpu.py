"""GPU: the CLI entry points mirror the reference's flags and stdout contract
(cluster_gcn/cluster_gcn.py:132-136,145-180; cluster_gcn_ist_distrib.py:475-479,520-564),
and the torch.distributed (RCCL) collectives used by the IST sync work on device tensors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TAIL = ['Training Time', 'Last Val', 'Best Val', 'Last Test', 'Best Test']


def test_cluster_gcn_cli_contract():
    from gist_amd import datasets
    from gist_amd.scripts import cluster_gcn as cli
    args = cli.build_parser().parse_args(
        ['--dataset', 'toy', '--n-epochs', '3', '--batch-size', '4', '--n-hidden', '32',
         '--n-layers', '2', '--lr', '0.01', '--use-layernorm', '--rnd-seed', '0', '--dropout', '0.2'])
    # reference defaults preserved
    d = cli.build_parser().parse_args([])
    assert (d.dropout, d.lr, d.n_epochs, d.batch_size, d.psize, d.n_hidden, d.n_layers,
            d.rnd_seed, d.weight_decay) == (0.2, 3e-2, 40, 20, 1500, 128, 1, 3, 0)
    lines = []
    res = cli.main(args, dataset=datasets.toy(), log=lambda *a, **k: lines.append(' '.join(map(str, a))))
    assert [l.split(':')[0] for l in lines[-5:]] == TAIL
    for l in lines[-5:]:
        float(l.split(':')[1])                       # parseable like the sweep scripts do
    assert len(res['val_accs']) == 3 and all(0 <= a <= 1 for a in res['val_accs'])
    assert res['total_time'] > 0
    # the nn.Module handed back shares storage with the trained arena
    W0 = res['model'].layers[0].linear.weight
    assert W0.is_cuda and torch.isfinite(W0).all()


def test_cluster_gcn_cli_module_host_path_trains_like_the_engine_path():
    """--host-path module: cluster_gcn.py's loop body statement for statement on the drop-in classes.  Same seeds, same
    construction order (ClusterIter, then the model), dropout 0: the same training as the engine path -- the five result
    lines, the accuracies of every epoch (two evaluators: to the last digit but one) and the trained weights."""
    from gist_amd import datasets
    from gist_amd.scripts import cluster_gcn as cli
    out = {}
    for hp in ('engine', 'module'):
        args = cli.build_parser().parse_args(
            ['--dataset', 'toy', '--n-epochs', '3', '--batch-size', '4', '--n-hidden', '32', '--n-layers', '2', '--lr', '0.01',
             '--use-layernorm', '--rnd-seed', '0', '--dropout', '0.0', '--host-path', hp])
        lines = []
        out[hp] = (cli.main(args, dataset=datasets.toy(), log=lambda *a, **k: lines.append(' '.join(map(str, a)))), lines)
        assert [l.split(':')[0] for l in lines[-5:]] == TAIL
    e, m = out['engine'][0], out['module'][0]
    assert getattr(m['model'], '_module_engines', None)          # (the module bound to the fused step)
    assert np.allclose(e['val_accs'], m['val_accs'], atol=2e-3) and np.allclose(e['test_accs'], m['test_accs'], atol=2e-3)
    for le, lm in zip(e['model'].layers, m['model'].layers):
        assert torch.equal(le.linear.weight, lm.linear.weight) and torch.equal(le.linear.bias, lm.linear.bias)


def test_ist_cli_world1_and_rccl_collectives():
    import torch.distributed as dist
    from gist_amd import datasets, ist
    from gist_amd.scripts import cluster_gcn_ist_distrib as cli
    d = cli.build_parser().parse_args([])
    assert (d.iter_per_site, d.num_subnet, d.dropout, d.lr, d.n_epochs, d.n_hidden, d.n_layers,
            d.weight_decay, d.dist_backend, d.dist_url, d.batch_size, d.psize, d.rnd_seed) == (
        5, 2, 0.5, 0.01, 20, 16, 1, 5e-4, 'nccl', 'tcp://127.0.0.1:9971', 20, 1500, 3)
    assert cli.build_parser().parse_args(['--use_layernorm', 'False']).use_layernorm is True  # quirk :538
    args = cli.build_parser().parse_args(
        ['--dataset', 'toy', '--num_subnet', '1', '--n-epochs', '2', '--batch-size', '4',
         '--n-hidden', '32', '--n-layers', '2', '--iter_per_site', '3', '--use_layernorm', 'True',
         '--dropout', '0.0', '--weight-decay', '0', '--dist-url', 'tcp://127.0.0.1:29877'])
    lines = []
    res = cli.main(args, dataset=datasets.toy(), log=lambda *a, **k: lines.append(' '.join(map(str, a))))
    assert [l.split(':')[0] for l in lines[-5:]] == TAIL
    assert res['events'].count('sync') >= 2
    # RCCL path of the collectives the sync uses, on device tensors
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29878', rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        comm = ist.TorchDistComm()
        inp = torch.arange(1000, dtype=torch.float32, device='cuda')
        out = torch.zeros(1000, device='cuda')
        comm.all_gather_flat(out, inp)
        comm.broadcast(inp, src=0)
        comm.barrier()
        torch.cuda.synchronize()
        assert torch.equal(out, inp)
    finally:
        dist.destroy_process_group()


def test_zero_edit_drop_in_script_matches_engine_path():
    """INTEGRATION.md route A: a script shaped like cluster_gcn/cluster_gcn.py main() --
    `import dgl` resolving to gist_amd.dgl_compat, ClusterIter yielding graph objects,
    `model(cluster)`, `loss.backward()`, `optimizer.step()`, `evaluate(...)` -- trains to the
    same parameters as the SageEngine fast path on the same data (dropout 0, same seeds)."""
    import random
    import torch.nn.functional as F
    import gist_amd.dgl_compat as dgl_compat
    dgl_compat.install()
    import dgl                                             # noqa: F401  (the alias)
    import dgl.function as fn                              # noqa: F401
    assert dgl is dgl_compat and fn.copy_src is dgl_compat.function.copy_src
    from gist_amd import datasets
    from gist_amd.modules import GCN
    from gist_amd.nn import CrossEntropyLoss
    from gist_amd.optim import Adam
    from gist_amd.sampler import ClusterIter
    from gist_amd.trainer import ClusterGCNTrainer
    from gist_amd.utils import evaluate

    dev = torch.device('cuda', 0)
    data = datasets.toy(seed=7)
    g = data.g
    train_nid = np.nonzero(g.ndata['train_mask'].numpy())[0].astype(np.int64)
    in_feats, n_classes = g.ndata['feat'].shape[1], data.num_classes
    psize, bs, hidden, L, lr = len(data.par_li), 4, 32, 2, 0.01

    # ---- reference-shaped script (module path) -------------------------------------------
    torch.manual_seed(5)
    random.seed(5)
    cluster_iterator = ClusterIter('toy', g, psize, bs, train_nid, use_pp=False,
                                   par_li=[p.copy() for p in data.par_li], device=dev)
    model = GCN(in_feats, hidden, n_classes, L, F.relu, 0.0, True, False, False, 1, True)
    init = [(l.linear.weight.detach().clone(), l.linear.bias.detach().clone()) for l in model.layers]
    model.cuda()
    loss_f = CrossEntropyLoss()
    optimizer = Adam(model.parameters(), lr=lr, weight_decay=0)
    g_dev = g.to(dev)
    for epoch in range(2):
        for j, cluster in enumerate(cluster_iterator):
            model.train()
            pred = model(cluster)
            batch_labels = cluster.ndata['label']
            batch_train_mask = cluster.ndata['train_mask']
            loss = loss_f(pred[batch_train_mask], batch_labels[batch_train_mask])
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
    acc_script = evaluate(model, g_dev, g_dev.ndata['label'], g_dev.ndata['val_mask'])
    assert torch.isfinite(loss) and 0.0 <= acc_script <= 1.0

    # ---- engine path, same seeds / data / init -----------------------------------------------
    random.seed(5)
    tr = ClusterGCNTrainer('toy', g, [p.copy() for p in data.par_li], psize, bs, hidden, L,
                           n_classes, 0.0, True, lr, 0.0, dev, init_params=init)
    for epoch in range(2):
        tr.train_epoch()
    for layer, (W, b) in zip(model.layers, tr.engine.arena.export()):
        assert np.abs(layer.linear.weight.detach().cpu().numpy() - W).max() < 1e-4
        assert np.abs(layer.linear.bias.detach().cpu().numpy() - b).max() < 1e-4
    assert abs(tr.evaluate('val_mask') - acc_script) < 1e-9


def test_ultra_wide_cli_and_save_results(tmp_path):
    """cluster_gcn_ist_ultra_wide.py's flags are cluster_gcn_ist_distrib.py's; here it is the same
    wrapper and loop with the row-blocked evaluator.  --save_results writes the pickle the
    reference's sweep drivers read ({fig_name}_result.pckl: total_time, trn_losses, val_accs,
    test_accs; cluster_gcn_ist_distrib.py:463-473) instead of the five lines; --use-pp is
    rejected loudly (it cannot work with ISTSAGELayer in the reference either)."""
    import pickle
    from gist_amd import datasets
    from gist_amd.scripts import cluster_gcn_ist_distrib as ref_cli
    from gist_amd.scripts import cluster_gcn_ist_ultra_wide as cli
    assert vars(cli.build_parser().parse_args([])) == vars(ref_cli.build_parser().parse_args([]))
    argv = ['--dataset', 'toy', '--num_subnet', '1', '--n-epochs', '2', '--batch-size', '4',
            '--n-hidden', '64', '--n-layers', '2', '--iter_per_site', '3', '--use_layernorm', 'True',
            '--dropout', '0.0', '--weight-decay', '0', '--fig-dir', str(tmp_path), '--fig-name', 'uw']
    lines = []
    log = lambda *a, **k: lines.append(' '.join(map(str, a)))
    res = cli.main(cli.build_parser().parse_args(argv + ['--dist-url', 'tcp://127.0.0.1:29879']),
                   dataset=datasets.toy(), log=log)
    assert [l.split(':')[0] for l in lines[-5:]] == TAIL
    assert len(res['trn_losses']) == len(res['val_accs']) and all(l > 0 for l in res['trn_losses'])
    lines.clear()
    res2 = cli.main(cli.build_parser().parse_args(argv + ['--save_results', '--dist-url',
                                                          'tcp://127.0.0.1:29880']),
                    dataset=datasets.toy(), log=log)
    assert not any(l.startswith('Last Val') for l in lines)          # pickle instead of the lines
    got = pickle.load(open(tmp_path / 'uw_result.pckl', 'rb'))
    assert sorted(got) == ['test_accs', 'total_time', 'trn_losses', 'val_accs']
    assert got['val_accs'] == res2['val_accs'] and got['trn_losses'] == res2['trn_losses']
    with pytest.raises(NotImplementedError):
        cli.main(cli.build_parser().parse_args(argv + ['--use-pp']), dataset=datasets.toy(), log=log)


def test_evaluator_row_blocks_agree():
    """FullGraphEvaluator evaluates a layer block of rows by block of rows (what lets the
    ultra-wide model evaluate in HBM): any row block gives the same logits as one block."""
    from gist_amd import datasets
    from gist_amd.engine import ParamArena, dims_for
    from gist_amd.trainer import FullGraphEvaluator
    dev = torch.device('cuda', 0)
    ds = datasets.toy(seed=3)
    g = ds.g
    dims = dims_for(g.ndata['feat'].shape[1], 96, ds.num_classes, 3)
    arena = ParamArena(dims, dev, with_grads=False)
    gen = torch.Generator().manual_seed(2)
    arena.load([((torch.rand(o, 2 * i, generator=gen) - 0.5) * 0.4, (torch.rand(o, generator=gen) - 0.5) * 0.4)
                for (i, o) in dims])
    whole = FullGraphEvaluator(g, dims, True, arena, dev, row_block=g.number_of_nodes())
    ref = whole.forward().clone()
    for rb in (1000, 257, 64):
        ev = FullGraphEvaluator(g, dims, True, arena, dev, row_block=rb)
        assert ev.row_block == rb
        assert (ev.forward() - ref).abs().max().item() < 1e-5
        assert abs(ev.accuracy('val_mask') - whole.accuracy('val_mask')) < 1e-9


@pytest.mark.parametrize('hidden', [256, 2048])
def test_evaluator_block_diagonal_split_agrees(hidden):
    """FullGraphEvaluator(node_blocks=...): A = A_diag + A_rest (inside the blocks on the blocked kernels --
    the LDS gather at width 256, counts x features on the matrix cores at 2048 -- the remainder gathered and
    accumulated, column tile by column tile) gives the logits of the one-pass aggregation, for row blocks that
    end at block boundaries; blocks of unequal sizes up to 128 nodes, a graph whose ids are ordered by block."""
    from gist_amd import datasets
    from gist_amd.engine import ParamArena, dims_for
    from gist_amd.trainer import FullGraphEvaluator
    dev = torch.device('cuda', 0)
    ds = datasets.make_block_dataset('blocks', 5000, 47, 64, 5, intra_deg=20, inter_deg=6, seed=4, hub_frac=0.01,
                                     hub_mult=8, train_frac=0.7)
    g = ds.g
    sizes = np.array([len(b) for b in np.array_split(np.arange(5000), 47)])
    bounds = np.concatenate([[0], np.cumsum(sizes)])
    assert sizes.max() <= 128
    dims = dims_for(64, hidden, 5, 3)
    arena = ParamArena(dims, dev, with_grads=False)
    gen = torch.Generator().manual_seed(2)
    arena.load([((torch.rand(o, 2 * i, generator=gen) - 0.5) * (2.0 / np.sqrt(2 * i)),
                 (torch.rand(o, generator=gen) - 0.5) * 0.1) for (i, o) in dims])
    whole = FullGraphEvaluator(g, dims, True, arena, dev, node_blocks=False)
    assert whole.split is None
    ref = whole.forward().clone()
    assert FullGraphEvaluator(g, dims, True, arena, dev).split is not None      # the dataset's own boundaries
    for rb in (5000, 1300):
        ev = FullGraphEvaluator(g, dims, True, arena, dev, row_block=rb, node_blocks=bounds)
        assert ev.split is not None and ev.split['diag_edges'] + ev.split['rest_edges'] == g.number_of_edges()
        assert ev.split['diag_edges'] > ev.split['rest_edges']
        assert all(c in set(bounds.tolist()) for c in ev.row_cuts) and ev.row_cuts[-1] == 5000
        if rb == 1300:
            assert len(ev.row_cuts) > 3
        assert (ev.forward() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
        assert abs(ev.accuracy('val_mask') - whole.accuracy('val_mask')) < 1e-9
    with pytest.raises(ValueError):
        FullGraphEvaluator(g, dims, True, arena, dev, node_blocks=np.array([0, 200, 5000]))      # blocks > 128


def test_block_units_kernel_against_float64():
    """gist_spmm_block_units_f32: per unit y[r0:r1] (+)= scale * C_u @ x[xs0:xs1] on the bf16x3 matrix cores, on unequal
    blocks (1 .. 128 rows / sources), output rows relative to a window of y, accumulate on and off, the largest exact
    count (256)."""
    from gist_amd import hip, _lib
    dev = torch.device('cuda', 0)
    rs = np.random.RandomState(0)
    sizes = np.array([100, 128, 1, 57, 128, 90])
    bounds = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(bounds[-1])
    stride = int(_lib.load().gist_spmm_block_image_bytes()) // 2
    assert stride >= 16384
    # one launch = pairs with disjoint output rows: (row block, column block)
    launches = [[(0, 1), (1, 0), (2, 4), (4, 0), (5, 3)], [(0, 3), (1, 5), (4, 1)], [(4, 2)]]
    scale = rs.rand(n).astype(np.float32) + 0.5
    t32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for d in (512, 128, 260):
        x = rs.randn(n, d + 4).astype(np.float32)
        lo, hi = 0, n                                           # y window = all rows, then a sub-window below
        for win in ((0, n), (int(bounds[1]), int(bounds[5]))):
            y0 = rs.randn(win[1] - win[0], d).astype(np.float32)
            y = t32(y0.copy())
            ref = y0.astype(np.float64)
            for pairs in launches:
                pairs = [(rb, cb) for rb, cb in pairs if bounds[rb] >= win[0] and bounds[rb + 1] <= win[1]]
                if not pairs:
                    continue
                imgs = np.zeros((len(pairs), stride), np.float32)
                units = np.zeros((len(pairs), 4), np.int32)
                for u, (rb, cb) in enumerate(pairs):
                    c = rs.poisson(0.4, (sizes[rb], sizes[cb])).astype(np.float64)
                    c[rs.randint(0, sizes[rb]), rs.randint(0, sizes[cb])] = 256
                    im = np.zeros((16, 128, 8), np.float32)
                    for k in range(sizes[cb]):
                        im[k // 8, :sizes[rb], k % 8] = c[:, k]
                    imgs[u, :16384] = im.ravel()
                    units[u] = (bounds[rb] - win[0], bounds[rb + 1] - win[0], bounds[cb], bounds[cb + 1])
                    ref[bounds[rb] - win[0]:bounds[rb + 1] - win[0]] += \
                        (c @ x[bounds[cb]:bounds[cb + 1], :d].astype(np.float64)) * scale[bounds[rb]:bounds[rb + 1], None]
                hip.spmm_block_units(t32(units), t32(imgs).to(torch.bfloat16).contiguous(), t32(x)[:, :d], y,
                                     out_scale=t32(scale[win[0]:win[1]].copy()), accumulate=True)
            assert np.abs(y.cpu().numpy() - ref).max() < 2e-5 * max(1.0, np.abs(ref).max()), (d, win)
        # accumulate = 0 overwrites the unit's rows
        rb, cb = 3, 4
        c = rs.poisson(0.4, (sizes[rb], sizes[cb])).astype(np.float64)
        im = np.zeros((16, 128, 8), np.float32)
        for k in range(sizes[cb]):
            im[k // 8, :sizes[rb], k % 8] = c[:, k]
        img = np.zeros((1, stride), np.float32)
        img[0, :16384] = im.ravel()
        y = torch.full((n, d), 7.0, device=dev)
        hip.spmm_block_units(t32(np.array([[bounds[rb], bounds[rb + 1], bounds[cb], bounds[cb + 1]]], np.int32)),
                             t32(img).to(torch.bfloat16).contiguous(), t32(x)[:, :d], y, accumulate=False)
        want = c @ x[bounds[cb]:bounds[cb + 1], :d].astype(np.float64)
        got = y.cpu().numpy()
        assert np.abs(got[bounds[rb]:bounds[rb + 1]] - want).max() < 2e-5 * max(1.0, np.abs(want).max())
        assert np.all(got[:bounds[rb]] == 7.0) and np.all(got[bounds[rb + 1]:] == 7.0)


def test_block_chains_kernel_against_float64():
    """gist_spmm_block_chains_f32 (round 5): chain c = several units with the SAME output rows -- a row block's diagonal
    block and its dense off-diagonal pairs -- summed in one workgroup's registers, y read / written once:
    y[r0:r1] (+)= scale * sum_u C_u @ x[xs0_u:xs1_u].  Chains of 1 .. 5 units over unequal blocks (1 .. 128 rows / sources),
    an EMPTY chain, widths that are not a multiple of the 128-column tile, output rows relative to a window, accumulate on
    and off, the largest exact count (256); against float64 and against the same units through gist_spmm_block_units_f32."""
    from gist_amd import hip, _lib
    dev = torch.device('cuda', 0)
    rs = np.random.RandomState(1)
    sizes = np.array([100, 128, 1, 57, 128, 90, 33])
    bounds = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(bounds[-1])
    stride = int(_lib.load().gist_spmm_block_image_bytes()) // 2
    # chains: row block -> its source blocks (the first is its own diagonal block)
    chains = {0: [0, 1, 3], 1: [1], 2: [2, 4, 0, 5, 6], 3: [], 4: [4, 0], 5: [5, 3, 1, 2], 6: [6, 6]}
    scale = rs.rand(n).astype(np.float32) + 0.5
    t32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for d in (512, 132, 1024 + 260):
        x = rs.randn(n, d + 4).astype(np.float32)
        for win, acc in (((0, n), True), ((0, n), False), ((int(bounds[1]), int(bounds[6])), True)):
            rows = [rb for rb in chains if bounds[rb] >= win[0] and bounds[rb + 1] <= win[1]]
            y0 = rs.randn(win[1] - win[0], d).astype(np.float32)
            ref = y0.astype(np.float64) if acc else np.full((win[1] - win[0], d), np.nan)
            units, imgs, cptr = [], [], [0]
            for rb in rows:
                tot = np.zeros((sizes[rb], d))
                for cb in chains[rb]:
                    c = rs.poisson(0.4, (sizes[rb], sizes[cb])).astype(np.float64)
                    c[rs.randint(0, sizes[rb]), rs.randint(0, sizes[cb])] = 256
                    im = np.zeros((16, 128, 8), np.float32)
                    for k in range(sizes[cb]):
                        im[k // 8, :sizes[rb], k % 8] = c[:, k]
                    img = np.zeros(stride, np.float32)
                    img[:16384] = im.ravel()
                    imgs.append(img)
                    units.append((bounds[rb] - win[0], bounds[rb + 1] - win[0], bounds[cb], bounds[cb + 1]))
                    tot += c @ x[bounds[cb]:bounds[cb + 1], :d].astype(np.float64)
                cptr.append(len(units))
                if chains[rb]:
                    lo, hi = bounds[rb] - win[0], bounds[rb + 1] - win[0]
                    ref[lo:hi] = (ref[lo:hi] if acc else 0.0) + tot * scale[bounds[rb]:bounds[rb + 1], None]
            U = t32(np.array(units, np.int32).reshape(-1, 4))
            I = t32(np.stack(imgs)).to(torch.bfloat16).contiguous()
            y = t32(y0.copy())
            hip.spmm_block_chains(t32(np.array(cptr, np.int32)), U, I, t32(x)[:, :d], y,
                                  out_scale=t32(scale[win[0]:win[1]].copy()), accumulate=acc)
            got = y.cpu().numpy()
            touched = ~np.isnan(ref[:, 0])
            assert np.abs(got[touched] - ref[touched]).max() < 2e-5 * max(1.0, np.abs(ref[touched]).max()), (d, win, acc)
            assert np.array_equal(got[~touched], y0[~touched])            # rows of empty / absent chains: untouched
            if acc:      # the same units one by one through the per-unit kernel: equal to rounding
                y2 = t32(y0.copy())
                for u in range(U.shape[0]):
                    hip.spmm_block_units(U[u:u + 1], I[u:u + 1], t32(x)[:, :d], y2,
                                         out_scale=t32(scale[win[0]:win[1]].copy()), accumulate=True)
                assert (y2 - y).abs().max().item() < 2e-5 * max(1.0, np.abs(ref[touched]).max())


@pytest.mark.parametrize('locality', [0.8, 1.0])
def test_evaluator_dense_block_pairs_agree(locality):
    """A graph whose inter-part edges go mostly to a few neighbour parts (what a partition of a real graph looks like):
    the evaluator finds the dense off-diagonal block pairs, runs them as counts x features, gathers the rest, and the
    logits equal the one-pass evaluator's; with the threshold out of reach no pair is dense and nothing changes."""
    from gist_amd import datasets
    from gist_amd.engine import ParamArena, dims_for
    from gist_amd.trainer import FullGraphEvaluator
    dev = torch.device('cuda', 0)
    ds = datasets.make_block_dataset('blocks', 6000, 60, 64, 5, intra_deg=20, inter_deg=24, seed=4, hub_frac=0.01,
                                     hub_mult=8, train_frac=0.7, inter_locality=locality)
    g = ds.g
    dims = dims_for(64, 256, 5, 3)
    arena = ParamArena(dims, dev, with_grads=False)
    gen = torch.Generator().manual_seed(2)
    arena.load([((torch.rand(o, 2 * i, generator=gen) - 0.5) * (2.0 / np.sqrt(2 * i)),
                 (torch.rand(o, generator=gen) - 0.5) * 0.1) for (i, o) in dims])
    ref = FullGraphEvaluator(g, dims, True, arena, dev, node_blocks=False).forward().clone()
    for rb in (6000, 1300):
        ev = FullGraphEvaluator(g, dims, True, arena, dev, row_block=rb, pair_min_edges=200)
        sp = ev.split
        assert sp['n_pairs'] >= 60 * 8 * 0.9 and sp['pair_edges'] > 0
        assert sp['diag_edges'] + sp['pair_edges'] + sp['rest_edges'] == g.number_of_edges()
        if locality == 1.0:
            assert sp['rest_edges'] < 0.02 * g.number_of_edges()        # (hub rows' thin pairs only)
        assert (ev.forward() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
        assert sp.get('chains') is not None                          # (round 5: diagonal block + pairs as chains)
    import os
    os.environ['GIST_EVAL_CHAINS'] = '0'                             # round 4's form: one launch per pair rank
    try:
        ev1 = FullGraphEvaluator(g, dims, True, arena, dev, pair_min_edges=200)
        assert ev1.split.get('chains') is None and ev1.split['n_pairs'] > 0
        assert (ev1.forward() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    finally:
        del os.environ['GIST_EVAL_CHAINS']
    ev0 = FullGraphEvaluator(g, dims, True, arena, dev, pair_min_edges=10 ** 9)
    assert ev0.split['n_pairs'] == 0
    assert (ev0.forward() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
