'''
traversal stack

The reference simulates a per-thread integer stack in global memory (GlobalStack,
reference stack.py:10-60: val[512*512][32] + len[], three global accesses per push/pop).
Here the stack is a per-lane LIFO in LDS laid out [level][lane] (csrc/pt_device.h `Stack`),
sized from the built tree's depth, so this module only keeps the names scripts import.
'''

from .common import Singleton, register


@register
class GlobalStack(metaclass=Singleton):
    def __init__(self, N_mt=512 * 512, N_len=32):
        self.N_mt = N_mt
        self.N_len = N_len


LocalStack = GlobalStack


def Stack():
    return GlobalStack()
