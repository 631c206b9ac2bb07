"""CPU: the host-side pieces of the drop-in module path that need no GPU (gist_amd/graph.py AllRowsMask,
gist_amd/module_engine.py StepLoss / eligibility): what they do to the reference's statements
`pred[batch_train_mask]`, `batch_labels[batch_train_mask]` and `loss.backward()` (cluster_gcn/cluster_gcn.py:98-104)."""
import torch
import torch.nn.functional as F


def test_all_rows_mask_indexes_without_a_gather_and_is_an_ordinary_mask_otherwise():
    from gist_amd.graph import AllRowsMask
    m = torch.ones(6, dtype=torch.bool).as_subclass(AllRowsMask)
    pred = torch.randn(6, 3, requires_grad=True)
    labels = torch.arange(6)
    assert pred[m] is pred and labels[m] is labels          # x[mask] with mask.numel() rows: x itself (no nonzero())
    assert torch.equal(pred[:, 1][m], pred[:, 1])
    short = torch.randn(4, 3)
    try:                                                   # another length: torch's own indexing (and its error)
        short[m]
        raise AssertionError('expected an IndexError')
    except IndexError:
        pass
    # everything else: a bool tensor
    assert m.dtype == torch.bool and bool(m.all()) and int(m.sum()) == 6 and (~m).sum().item() == 0
    assert type(m & torch.tensor([True, False] * 3)) is torch.Tensor
    assert torch.equal(torch.arange(6)[m.clone()], torch.arange(6))
    # the gradient flows through the identity
    F.cross_entropy(pred[m], labels[m] % 3).backward()
    assert pred.grad is not None and pred.grad.shape == pred.shape


def test_step_loss_without_a_pending_step_is_an_ordinary_tensor():
    from gist_amd.module_engine import StepLoss
    w = torch.randn(5, requires_grad=True)
    loss = (w * w).sum().as_subclass(StepLoss)
    assert type(loss * 2) is torch.Tensor                   # results of operations are plain tensors
    loss.backward()                                        # no fused step behind it: torch's own backward
    assert torch.allclose(w.grad, 2 * w.detach())
    assert abs(float(loss) - float((w * w).sum())) < 1e-6


def test_eligibility_of_a_model_for_the_step_plan():
    from gist_amd.modules import GCN, GraphSAGELayer
    from gist_amd.module_engine import eligible
    assert eligible(GCN(10, 16, 3, 2, F.relu, 0.2, True, False, False, 1, True))
    assert eligible(GCN(10, 16, 3, 2, F.relu, 0.0, False, False, True, 2, True))      # a sub-GCN, no LayerNorm
    assert not eligible(GCN(10, 16, 3, 2, torch.tanh, 0.2, True, False, False, 1, True))   # another activation
    m = GCN(10, 16, 3, 2, F.relu, 0.2, True, False, False, 1, True)
    m.layers[1] = GraphSAGELayer(16, 16, F.relu, 0.2)       # another layer type
    assert not eligible(m)
    m = GCN(10, 16, 3, 2, F.relu, 0.2, True, False, False, 1, True)
    m.layers[1].p_drop = 0.5                                # layers that disagree on the dropout probability
    assert not eligible(m)
