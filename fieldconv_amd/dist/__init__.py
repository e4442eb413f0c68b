from .halo import HaloPlan, halo_exchange

__all__ = ['HaloPlan', 'halo_exchange']
