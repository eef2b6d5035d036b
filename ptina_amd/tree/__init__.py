'''
acceleration structure selector (reference tree/__init__.py:5-6 picks lbvh.BVHTree)
'''

from .lbvh import *                   # noqa: F401,F403
