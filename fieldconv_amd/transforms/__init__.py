from .fc_precomp import FCPrecomp

__all__ = ['FCPrecomp']
