from .synthetic import random_support, sphere_partition, sphere_support

__all__ = ['random_support', 'sphere_support', 'sphere_partition']
