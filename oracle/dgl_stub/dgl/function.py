"""Stub of dgl.function: message/reduce builtins are returned as tags."""


def copy_src(src, out):
    return ('copy_src', src, out)


copy_u = copy_src


def sum(msg, out):  # noqa: A001 - mirrors dgl.function.sum
    return ('sum', msg, out)
