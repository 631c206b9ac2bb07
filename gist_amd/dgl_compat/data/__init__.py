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
    from ...datasets import load_dataset
    return load_dataset(args.dataset, getattr(args, 'data_root', None))


class DGLDataset(object):
    def __init__(self, name=None, save_dir=None, force_reload=False, verbose=False):
        self.name = name


class PPIDataset(object):
    def __init__(self, mode):
        raise NotImplementedError('gist_amd: PPI dataset is out of scope')
