"""dgl.data names the in-scope scripts import (cluster_gcn_ist_distrib.py:8, utils.py:6-7)."""
from . import utils  # noqa: F401


def register_data_args(parser):
    parser.add_argument('--dataset', type=str, required=False, default='reddit-synth',
                        help='dataset name (synthetic generators: reddit-synth, amazon-synth, toy)')


def load_data(args):
    from ...datasets import load_dataset
    return load_dataset(args.dataset)


class DGLDataset(object):
    def __init__(self, name=None, save_dir=None, force_reload=False, verbose=False):
        self.name = name


class PPIDataset(object):
    def __init__(self, mode):
        raise NotImplementedError('gist_amd: PPI dataset is out of scope')
