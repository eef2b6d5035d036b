'''
common utilities used by this package (host side only: the device math of the reference's
common.py lives in csrc/pt_device.h)
'''

import numpy as np

from . import _lib

eps = 1e-6        # reference common.py:32
inf = 1e6         # reference common.py:33


class Singleton(type):
    '''Foo() always returns the one instance; only the first call's arguments count
    (reference common.py:407-413)'''
    _instance = None

    def __call__(cls, *args, **kwargs):
        if cls._instance is None:
            cls._instance = super().__call__(*args, **kwargs)
        return cls._instance


_singletons = []


def register(cls):
    _singletons.append(cls)
    return cls


def reset_all():
    '''drop every singleton and the device context (tests; the reference has no equivalent
    because a Taichi program cannot be re-initialised)'''
    _lib.drop_context()          # first: a launch in flight may still be writing into an array a singleton holds (FilmTable._next)
    for cls in _singletons:
        cls._instance = None


def ctx():
    return _lib.get_context()
