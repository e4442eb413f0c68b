from .field import EPS, isZero, isOrigin, softAbs, softAngle, softAbsolute, softSqrt

__all__ = ['EPS', 'isZero', 'isOrigin', 'softAbs', 'softAngle', 'softAbsolute', 'softSqrt']
