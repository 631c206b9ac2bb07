"""Stub of dgl.data: only the names the reference imports."""
from . import utils  # noqa: F401


def register_data_args(parser):
    parser.add_argument('--dataset', type=str, required=False, default='synthetic')


def load_data(args):
    raise RuntimeError('no datasets offline')


class DGLDataset(object):
    def __init__(self, name=None, save_dir=None, force_reload=False, verbose=False):
        self.name = name


class PPIDataset(object):
    def __init__(self, mode):
        raise RuntimeError('no datasets offline')
