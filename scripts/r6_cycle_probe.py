"""Which reference cycles keep a dropped module-path model alive?  (gc.DEBUG_SAVEALL after one bound model's life)"""
import gc, random, collections
import numpy as np, torch, torch.nn.functional as F
from gist_amd import datasets
from gist_amd.modules import GCN
from gist_amd.nn import CrossEntropyLoss
from gist_amd.optim import Adam
from gist_amd.sampler import ClusterIter
DEV = torch.device('cuda:0')


def one():
    ds = datasets.toy(seed=9, n=3000, n_blocks=30, n_feats=100, n_classes=6, train_frac=1.0)
    g = ds.g
    random.seed(4)
    it = ClusterIter('toy', g, len(ds.par_li), 5, np.arange(g.number_of_nodes(), dtype=np.int64),
                     par_li=[p.copy() for p in ds.par_li], device=DEV)
    torch.manual_seed(3)
    model = GCN(100, 256, 6, 2, F.relu, 0.0, True, False, False, 1, True).cuda()
    loss_f = CrossEntropyLoss(); opt = Adam(model.parameters(), lr=0.01)
    model.train()
    for cluster in it:
        pred = model(cluster); loss = loss_f(pred, cluster.ndata['label'])
        opt.zero_grad(); loss.backward(); opt.step()
    float(loss)


one(); gc.collect(); torch.cuda.synchronize()
base = torch.cuda.memory_allocated()
gc.disable()
one()
torch.cuda.synchronize()
print('leaked by reference counting alone:', torch.cuda.memory_allocated() - base)
def who(o, depth=0, seen=None):
    seen = seen or set()
    if depth > 4 or id(o) in seen:
        return
    seen.add(id(o))
    for r in gc.get_referrers(o):
        if isinstance(r, (type(gc), type)) or r is globals() or type(r).__name__ in ('frame', 'list_iterator'):
            continue
        label = type(r).__name__
        if isinstance(r, dict):
            owners = [x for x in gc.get_referrers(r) if getattr(x, '__dict__', None) is r]
            if owners:
                keys = [str(k) for k, v in r.items() if v is o]
                print('  ' * depth, type(owners[0]).__name__, '.', keys)
                who(owners[0], depth + 1, seen)
                continue
        if isinstance(r, (tuple, list)) and len(r) > 100:
            continue
        print('  ' * depth, label, (len(r) if hasattr(r, '__len__') else ''))
        who(r, depth + 1, seen)


for o in gc.get_objects():
    if type(o).__name__ in ('SageEngine',):
        print('== SageEngine', id(o))
        who(o)
