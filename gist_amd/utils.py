"""Evaluation helpers with the reference's signatures (cluster_gcn/utils.py:47-80)."""
import torch

from . import hip


def evaluate(model, g, labels, mask, method='acc'):
    """Full-graph forward in eval mode + accuracy (utils.py:70-80).

    micro-F1 of a single-label argmax prediction equals accuracy (utils.py:47-56 with
    average='micro'), so both methods share one kernel."""
    assert method in ['acc', 'f1'], 'invalid method'
    model.eval()
    with torch.no_grad():
        logits = model(g)
        dev = logits.device
        lab = labels.to(dev).to(torch.int32).contiguous()
        msk = mask.to(dev).to(torch.uint8).contiguous()
        total = int(msk.sum().item())
        if total == 0:
            return -1
        correct = torch.zeros(1, dtype=torch.int32, device=dev)
        hip.argmax_correct(logits.contiguous(), lab, msk, correct)
        return correct.item() / total
