from .grad_sync import GradientBuckets
from .halo import HaloPlan, halo_exchange, overlap_backward, overlap_forward

__all__ = ['GradientBuckets', 'HaloPlan', 'halo_exchange', 'overlap_backward', 'overlap_forward']
