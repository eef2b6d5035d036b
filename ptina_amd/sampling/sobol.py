'''
Sobol quasi-random generator (reference sampling/sobol.py).

Direction numbers come from the Joe-Kuo table new-joe-kuo-6.21201
(https://web.maths.unsw.edu.au/~fkuo/sobol/), which the reference reads from the
un-vendored `pysobol` package; the same public table is shipped here as
data/joe_kuo_21201.npz (tools/make_joe_kuo_fixture.py).
'''

import os

from ..common import *                # noqa: F401,F403
from ..common import Singleton, register, ctx, np
from .._lib import fptr, iptr
import ctypes as C

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'data',
                     'joe_kuo_21201.npz')


def calc_sobol_vgrid(N, D):
    '''direction-number grid V[L+1, D], bit 31 = 2^-1 (reference sobol.py:32-70), computed
    for all dimensions of equal degree s at once'''
    z = np.load(_DATA)
    if D > z['s'].shape[0]:
        raise ValueError(f'at most {z["s"].shape[0]} Sobol dimensions are tabulated')
    s_all = z['s'][:D].astype(np.int64)
    a_all = z['a'][:D].astype(np.int64)
    m_all = z['m'][:D].astype(np.int64)

    L = int(np.ceil(np.log2(N)))
    V = np.zeros((L + 1, D), np.int64)

    # dimension 0: van der Corput, m_i = 1 (sobol.py:53-55)
    for i in range(L + 1):
        V[i, 0] = 1 << (32 - i)

    for s in np.unique(s_all[1:]):
        s = int(s)
        idx = np.nonzero(s_all == s)[0]
        idx = idx[idx != 0]
        m = m_all[idx]
        a = a_all[idx]
        for i in range(1, min(s, L) + 1):
            V[i, idx] = m[:, i - 1] << (32 - i)
        for i in range(s + 1, L + 1):
            v = V[i - s, idx] ^ (V[i - s, idx] >> s)
            for k in range(1, s):
                v ^= ((a >> (s - 1 - k)) & 1) * V[i - k, idx]
            V[i, idx] = v
    return V


@register
class SobolSampler(metaclass=Singleton):
    def __init__(self, dim=21201, nsamples=2**20, skip=64):
        self.dim = dim
        self.nsamples = nsamples
        self.skip = skip
        V = calc_sobol_vgrid(nsamples, dim)
        self._V = np.ascontiguousarray((V & 0xffffffff).astype(np.uint32).view(np.int32))
        ctx().call('mpt_sobol_init', iptr(self._V), self._V.shape[0], self._V.shape[1])
        self.reset()

    def reset(self):
        '''reference sobol.py:92-97: time = 0, X = 0, then `skip` updates'''
        ctx().call('mpt_sobol_reset', int(self.skip))

    def update(self):
        '''reference sobol.py:99-105'''
        ctx().call('mpt_sobol_update', 1)

    def state(self):
        X = np.zeros(self.dim, np.int32)
        P = np.zeros(self.dim, np.float32)
        t = C.c_int32(0)
        ctx().call('mpt_sobol_get', iptr(X), fptr(P), C.byref(t))
        return t.value, X, P

    @property
    def time(self):
        return self.state()[0]
