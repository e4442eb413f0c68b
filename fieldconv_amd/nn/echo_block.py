import torch.nn as nn
import torch.nn.functional as F

from .. import functional as _fn
from .echo import ECHO
from .field_conv import FieldConv
from .tangent_nonlin import TangentNonLin


def histDim(n_bins):
    """Number of raster cells inside the disk (reference nn/echo_block.py:10-18)."""
    w = 2 * n_bins + 1
    return sum(1 for i in range(w) for j in range(w)
               if (i - n_bins) ** 2 + (j - n_bins) ** 2 <= (n_bins + 0.25) ** 2)


class ECHOBlock(nn.Module):
    """FieldConv -> modReLU -> ECHO descriptors -> MLP, plus a linear residual on |x|
    (reference nn/echo_block.py:20-103).  Converts tangent features to scalar features."""

    def __init__(self, in_channels, out_channels, n_des=None, n_bins=3, band_limit=1, n_rings=6, ftype=1):
        super().__init__()
        if n_des is None:
            n_des = in_channels
        self.conv = FieldConv(in_channels, n_des, band_limit, n_rings, ftype)
        # the reference sizes this by in_channels although it acts on n_des channels
        # (nn/echo_block.py:57); kept for state_dict compatibility
        self.nonlin = TangentNonLin(in_channels)
        self.echo = ECHO(n_des, n_bins)
        mid = n_des * histDim(n_bins)
        self.lin1 = nn.Linear(mid, 128)
        self.lin2 = nn.Linear(128, 64)
        self.lin3 = nn.Linear(64, out_channels)
        self.res = nn.Linear(in_channels, out_channels)
        self.n_des = n_des

    def forward(self, x, supp_edges, supp_sten, ln, wxp):
        bias = self.nonlin.bias
        if bias.shape[1] < self.n_des:
            raise ValueError('ECHOBlock requires n_des <= in_channels (reference nn/echo_block.py:57,93)')
        # reference behaviour: bias[0, channel index] is gathered per entry, so only the first n_des biases are ever used;
        # the modReLU runs in the convolution's epilogue
        d = None
        if _fn.on_device(x) and supp_sten.dim() == 3 and supp_sten.shape[1] == self.conv.R and supp_sten.shape[2] == 2 * self.conv.B + 1:
            # convolution + modReLU + descriptors as one autograd node and one native call per pass (csrc/fc_blocks.hip)
            from ..blocks import echo_block_descriptors
            from ..graph import get_graph
            d = echo_block_descriptors(self, x, get_graph(supp_edges, supp_sten, x.shape[0]), ln, wxp)
        if d is None:
            h = self.conv.forward_act(x, supp_edges, supp_sten, bias[:, : self.n_des])
            d = self.echo(h, supp_edges, ln, wxp)
        d = d.reshape(d.shape[0], -1)
        if _fn.on_device(x):        # the MLP and the residual on |x| as one autograd node (the reference's arithmetic, written out)
            from ..blocks import echo_block_tail
            out = echo_block_tail(self, d, x)
            if out is not None:
                return out
        d = F.relu(self.lin1(d))
        d = F.relu(self.lin2(d))
        return self.lin3(d) + self.res(_fn.soft_abs(x))
