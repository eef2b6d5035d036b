#!/usr/bin/env python3
'''
Generate ptina_amd/data/joe_kuo_21201.npz and tests/golden/sobol_points.npz.

PTina reads the Joe-Kuo "new-joe-kuo-6.21201" stream from the un-vendored
third-party package `pysobol` (reference ptina/sampling/sobol.py:35,49-53) as a
flat list  s, a, m_1..m_s  per dimension d >= 2.  pysobol is not installed here,
but scipy ships the very same public table (as primitive polynomial + initial
direction numbers).  This script converts scipy's encoding to the (s, a, m)
triplets PTina consumes:

    s = bitlen(poly) - 1
    a = (poly >> 1) & (2**(s-1) - 1)        # interior coefficients
    m = vinit[:s]

Row j of the output is PTina's dimension j for j >= 1; row 0 is unused by PTina
(its dimension 0 is the hard-wired van der Corput sequence, sobol.py:53-55) and
is stored as s=0.

The golden points are an INDEPENDENT pin for the sampler: scipy's unscrambled
Sobol point k equals PTina's sampler state after k updates.
'''
import os
import numpy as np
import scipy
from scipy.stats import qmc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = 21201


def main():
    src = os.path.join(os.path.dirname(scipy.__file__), 'stats',
                       '_sobol_direction_numbers.npz')
    z = np.load(src)
    poly = z['poly'].astype(np.int64)
    vinit = z['vinit'].astype(np.int64)
    assert poly.shape == (D,) and vinit.shape == (D, 18)

    s = np.zeros(D, np.uint8)
    a = np.zeros(D, np.uint32)
    m = np.zeros((D, 18), np.uint32)
    for j in range(1, D):
        sj = int(poly[j]).bit_length() - 1
        s[j] = sj
        a[j] = (int(poly[j]) >> 1) & ((1 << (sj - 1)) - 1) if sj > 1 else 0
        m[j, :sj] = vinit[j, :sj]
        assert np.all(vinit[j, sj:] == 0)

    # published first rows of new-joe-kuo-6.21201 (d = 2, 3, 4, 8)
    assert (s[1], a[1], list(m[1, :1])) == (1, 0, [1])
    assert (s[2], a[2], list(m[2, :2])) == (2, 1, [1, 3])
    assert (s[3], a[3], list(m[3, :3])) == (3, 1, [1, 3, 1])
    assert (s[7], a[7], list(m[7, :5])) == (5, 2, [1, 1, 5, 5, 17])

    out = os.path.join(ROOT, 'ptina_amd', 'data', 'joe_kuo_21201.npz')
    np.savez_compressed(out, s=s, a=a, m=m)
    print('wrote', out, os.path.getsize(out), 'bytes')

    # golden points: scipy point k == PTina state after k updates
    K = 98
    pts = qmc.Sobol(d=D, scramble=False).random(K)
    ks = np.array([1, 2, 3, 64, 65, 66, 97], np.int32)
    P = pts[ks].astype(np.float32)
    assert np.array_equal(P.astype(np.float64), pts[ks])   # exact in f32
    out = os.path.join(ROOT, 'tests', 'golden', 'sobol_points.npz')
    np.savez_compressed(out, k=ks, P=P)
    print('wrote', out, os.path.getsize(out), 'bytes')


if __name__ == '__main__':
    main()
