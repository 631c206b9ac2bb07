"""dgl.function builtins used by the reference (cluster_gcn/modules.py:3,224-225)."""


class CopySrc(object):
    def __init__(self, src, out):
        self.src, self.out = src, out


class Sum(object):
    def __init__(self, msg, out):
        self.msg, self.out = msg, out


def copy_src(src, out):
    return CopySrc(src, out)


copy_u = copy_src


def sum(msg, out):  # noqa: A001 - mirrors dgl.function.sum
    return Sum(msg, out)
