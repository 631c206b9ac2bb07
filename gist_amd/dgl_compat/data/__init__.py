"""dgl.data names the in-scope scripts import (cluster_gcn_ist_distrib.py:8, utils.py:6-7)."""
from . import utils  # noqa: F401


def register_data_args(parser):
    parser.add_argument('--dataset', type=str, required=False, default='reddit-synth',
                        help='dataset name: reddit / reddit-self-loop / amazon2m read real files from '
                             '--data-root (or $GIST_DATA_ROOT) and fail if they are missing; '
                             'reddit-synth / amazon-synth / cora-synth / toy are seeded synthetic '
                             'stand-ins')
    parser.add_argument('--data-root', type=str, default=None,
                        help='directory with reddit_data.npz + reddit[_self_loop]_graph.npz, or '
                             'GraphSAGE-format {name}-G.json/-feats.npy/-id_map.json/-class_map.json')


def load_data(args):
    """dgl.data.load_data (cluster_gcn/utils.py:122; gcn/train.py:36): cluster datasets are
    Dataset tuples, the citation graphs of gcn/train.py legacy objects (`.features`, `.graph` ...)."""
    from ...datasets import load_dataset
    if args.dataset in ('cora', 'citeseer', 'pubmed'):
        raise FileNotFoundError(
            'gist_amd: the %s citation files are not available offline and there is no reader for '
            'them here; the seeded Cora-like stand-in is --dataset cora-synth' % args.dataset)
    return load_dataset(args.dataset, getattr(args, 'data_root', None))


class DGLDataset(object):
    def __init__(self, name=None, save_dir=None, force_reload=False, verbose=False):
        self.name = name


class PPIDataset(object):
    def __init__(self, mode):
        raise NotImplementedError('gist_amd: PPI dataset is out of scope')
