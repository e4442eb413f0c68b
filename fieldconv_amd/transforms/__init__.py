from .fc_precomp import FCPrecomp, radialInterpolant

__all__ = ['FCPrecomp', 'radialInterpolant']
