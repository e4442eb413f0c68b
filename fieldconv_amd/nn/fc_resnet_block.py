import torch.nn as nn

from .. import functional as _fn
from .field_conv import FieldConv
from .tangent_lin import TangentLin
from .tangent_nonlin import TangentNonLin


class FCResNetBlock(nn.Module):
    """nonlin2(res(x) + conv2(nonlin1(conv1(x)))) -- the FCResNet block of section 5 / figure 2
    (reference nn/fc_resnet_block.py:7-88).  `frontload` picks which convolution changes width."""

    def __init__(self, in_channels, out_channels, band_limit=1, n_rings=6, ftype=1, frontload=False):
        super().__init__()
        mid = in_channels if frontload else out_channels
        self.conv1 = FieldConv(in_channels, mid, band_limit=band_limit, n_rings=n_rings, ftype=ftype)
        self.conv2 = FieldConv(mid, out_channels, band_limit=band_limit, n_rings=n_rings, ftype=ftype)
        self.nonlin1 = TangentNonLin(mid)
        self.nonlin2 = TangentNonLin(out_channels)
        self.res = TangentLin(in_channels, out_channels)

    def forward(self, x, supp_edges, supp_sten):
        # the whole block as one autograd node and one native call per pass (csrc/fc_blocks.hip), where that applies ...
        if _fn.on_device(x) and supp_sten.dim() == 3 and supp_sten.shape[1] == self.conv1.R and supp_sten.shape[2] == 2 * self.conv1.B + 1:
            from ..blocks import resnet_block
            from ..graph import get_graph
            out = resnet_block(self, x, get_graph(supp_edges, supp_sten, x.shape[0]))
            if out is not None:
                return out
        # ... else nonlin1(conv1(x)) and nonlin2(res(x) + conv2(h)) with residual add and modReLU in the convolutions' epilogues
        h = self.conv1.forward_act(x, supp_edges, supp_sten, self.nonlin1.bias)
        return self.conv2.forward_act(h, supp_edges, supp_sten, self.nonlin2.bias, addend=self.res(x))
