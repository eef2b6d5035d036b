'''
sampling: Wang hash pixel decorrelation (reference sampling/__init__.py:9-23), host versions
(the device ones are in csrc/pt_device.h)
'''

import numpy as np


def wanghash(x):
    value = np.asarray(x).astype(np.uint32)
    with np.errstate(over='ignore'):
        value = (value ^ np.uint32(61)) ^ (value >> np.uint32(16))
        value = value * np.uint32(9)
        value = value ^ (value << np.uint32(4))
        value = value * np.uint32(0x27d4eb2d)
        value = value ^ (value >> np.uint32(15))
    return value.astype(np.int32)


def wanghash2(x, y):
    value = wanghash(x)
    return wanghash(np.asarray(y).astype(np.int32) ^ value)
