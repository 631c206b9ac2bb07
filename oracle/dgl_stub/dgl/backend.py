"""Stub of dgl.backend (partition_utils.py:8,16 uses asnumpy)."""


def asnumpy(t):
    return t.detach().cpu().numpy()
