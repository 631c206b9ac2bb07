"""TEST INFRASTRUCTURE (not product): the collectives of gist_amd.ist.TorchDistComm staged through
the host over `gloo`, for S rank processes that share ONE GPU on a 1-GPU box (the reference's own
launcher does the same with `--cuda-id 0`, script/reddit/run_ist_distrib.sh:16-18).  RCCL refuses
two ranks on one device, so the product's one collective cannot run there; everything around it
(HipBlocks gather/scatter, the replicated base, the schedule) is the product path.  Never used for
a reported number: bench.py marks such a run INVALID."""
import torch
import torch.distributed as dist

from gist_amd.ist import TorchDistComm


class HostStagedComm(TorchDistComm):
    def all_gather_flat(self, out, inp):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu(), group=self.group)
        out.copy_(o)

    def broadcast(self, t, src=0):
        c = t.cpu()
        dist.broadcast(c, src=src, group=self.group)
        t.copy_(c)
