"""Drop-in for the reference's code/networks/dsbn.py: DomainSpecificBatchNorm2d keeps one BatchNorm2d per
domain under ``bns`` (same state_dict keys ``bns.{d}.*``).  Inside Rec_Decoder / ConvU_Rec the normalisation runs
fused in the HIP conv kernels (one statistics group per domain); called on its own, ``forward(x, domain_label)``
normalises with ``bns[domain_label[0]]`` on the standalone HIP BatchNorm and returns ``(y, domain_label)`` like
dsbn.py:24-27."""
from torch import nn

from ramdsir.modules import FusedBatchNorm2d


class _DomainSpecificBatchNorm(nn.Module):
    _version = 2

    def __init__(self, num_features, num_domains, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super(_DomainSpecificBatchNorm, self).__init__()
        self.bns = nn.ModuleList(
            [FusedBatchNorm2d(num_features, eps, momentum, affine, track_running_stats) for _ in range(num_domains)])

    def reset_running_stats(self):
        for bn in self.bns:
            bn.reset_running_stats()

    def reset_parameters(self):
        for bn in self.bns:
            bn.reset_parameters()

    def _check_input_dim(self, input):
        raise NotImplementedError

    def forward(self, x, domain_label):
        self._check_input_dim(x)                       # dsbn.py:25: ValueError on non-4D input
        bn = self.bns[domain_label[0]]                 # dsbn.py:26: the first label picks the BatchNorm for the whole batch
        return bn(x), domain_label


class DomainSpecificBatchNorm2d(_DomainSpecificBatchNorm):
    def _check_input_dim(self, input):
        if input.dim() != 4:
            raise ValueError('expected 4D input (got {}D input)'.format(input.dim()))
