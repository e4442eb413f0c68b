from .field import EPS, isZero, isOrigin, softAbs, softAngle, softAbsolute, softSqrt

from .step_graph import StepGraph

__all__ = ['StepGraph', 'EPS', 'isZero', 'isOrigin', 'softAbs', 'softAngle', 'softAbsolute', 'softSqrt']
