"""dgl.backend.asnumpy (cluster_gcn/partition_utils.py:8,16)."""


def asnumpy(t):
    return t.detach().cpu().numpy()
