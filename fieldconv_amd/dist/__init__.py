from .halo import HaloPlan, halo_exchange, overlap_backward

__all__ = ['HaloPlan', 'halo_exchange', 'overlap_backward']
