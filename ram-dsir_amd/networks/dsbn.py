"""Drop-in for the reference's code/networks/dsbn.py.

``DomainSpecificBatchNorm2d`` owns one BatchNorm per domain in the ModuleList ``bns`` (the checkpoint keys are
``bns.{d}.weight`` ... ``bns.{d}.num_batches_tracked``).  Inside Rec_Decoder / ConvU_Rec the
normalisation is fused into the HIP conv kernels, one statistics group per domain; called on its own it runs the
standalone HIP BatchNorm of the domain the FIRST label names and hands the labels back, as dsbn.py:24-27 does."""
from torch import nn

from ramdsir.modules import FusedBatchNorm2d


class DomainSpecificBatchNorm2d(nn.Module):
    def __init__(self, num_features, num_domains, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bns = nn.ModuleList(FusedBatchNorm2d(num_features, eps, momentum, affine, track_running_stats)
                                 for _ in range(num_domains))

    def reset_running_stats(self):
        """Every domain's running_mean / running_var / num_batches_tracked back to 0 / 1 / 0 (dsbn.py:13-15)."""
        for d in range(len(self.bns)):
            self.bns[d].reset_running_stats()

    def reset_parameters(self):
        """Every domain's statistics AND affine parameters back to their initial values (dsbn.py:17-19)."""
        for d in range(len(self.bns)):
            self.bns[d].reset_parameters()

    def forward(self, x, domain_label):
        if x.dim() != 4:
            raise ValueError('expected 4D input (got {}D input)'.format(x.dim()))
        return self.bns[domain_label[0]](x), domain_label


# the reference's module also exports the dimension-agnostic base class (dsbn.py:4); here the 2-D class is the only one, under both names
_DomainSpecificBatchNorm = DomainSpecificBatchNorm2d
