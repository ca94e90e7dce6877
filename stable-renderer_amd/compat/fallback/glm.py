"""Fallback for scripts that ``import glm`` (PyGLM) on a machine without it: the handful of vector helpers the reference's example
scripts use in their own Component subclasses (scripts/boat_example.py: vec3, length, clamp), numpy backed.  ``compat.install()``
appends this directory at the END of sys.path, so a real PyGLM always wins."""
import math

import numpy as np


class vec3(np.ndarray):
    def __new__(cls, *a):
        if len(a) == 0:
            v = (0.0, 0.0, 0.0)
        elif len(a) == 1:
            v = (a[0],) * 3 if np.isscalar(a[0]) else tuple(a[0])
        else:
            v = a
        return np.asarray(v, np.float64).reshape(3).view(cls)

    x = property(lambda s: float(s[0]))
    y = property(lambda s: float(s[1]))
    z = property(lambda s: float(s[2]))


def length(v):
    return float(np.linalg.norm(np.asarray(v, np.float64)))


def normalize(v):
    v = np.asarray(v, np.float64)
    return (v / np.linalg.norm(v)).view(vec3)


def clamp(v, lo, hi):
    return np.clip(v, lo, hi)


def radians(d):
    return math.radians(d) if np.isscalar(d) else np.radians(d)


def degrees(r):
    return math.degrees(r) if np.isscalar(r) else np.degrees(r)


def dot(a, b):
    return float(np.dot(a, b))


def cross(a, b):
    return np.cross(a, b).view(vec3)
