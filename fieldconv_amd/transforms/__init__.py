from .fc_precomp import FCPrecomp
from .precomp_cache import load_precomp, save_precomp

__all__ = ['FCPrecomp', 'load_precomp', 'save_precomp']
