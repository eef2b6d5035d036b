import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ctypes as C
from ptina_amd import scenes, _lib
from ptina_amd.common import ctx, reset_all
from helpers import setup_engine
for name in ('s34', 's978'):
    reset_all()
    eng = setup_engine(scenes.get_scene(name), 16, 16, mode='fast')
    c = ctx()
    nw = C.c_int(0)
    c.call('mpt_get_wide', None, None, 0, C.byref(nw))
    w = np.zeros((nw.value, 8, 4), np.float32); q = np.zeros((nw.value, 4, 4), np.float32)
    c.call('mpt_get_wide', _lib.fptr(w), _lib.fptr(q), nw.value, C.byref(nw))
    lo = np.stack([w[:, 0], w[:, 2], w[:, 4]], axis=1)      # [node][axis][child]
    hi = np.stack([w[:, 1], w[:, 3], w[:, 5]], axis=1)
    unused = ~(lo[:, 0, :] < 1e29)
    pmax = max(np.abs(lo[lo < 1e29]).max(), np.abs(hi[hi < 1e29]).max()) * 1.0005
    # outward rounding to f16
    def rd(x):
        h = x.astype(np.float16); h = np.where(h.astype(np.float32) > x, np.nextafter(h, np.float16(-np.inf)), h); return h
    def ru(x):
        h = x.astype(np.float16); h = np.where(h.astype(np.float32) < x, np.nextafter(h, np.float16(np.inf)), h); return h
    lo16, hi16 = rd(lo), ru(hi)
    rng = np.random.default_rng(1)
    R = 20000
    o = rng.uniform([-2, 0, -2], [2, 4, 5], (R, 3)).astype(np.float32)
    d = rng.normal(size=(R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
    inv = (1.0 / d).astype(np.float32); oinv = (o * inv).astype(np.float32)
    tbest = rng.uniform(0.5, 8.0, R).astype(np.float32)
    bad = 0; extra = 0; total = 0
    for nd in range(nw.value):
        # f32 test
        neg = inv < 0
        pn = np.where(neg[:, :, None], hi[nd][None], lo[nd][None]); pf = np.where(neg[:, :, None], lo[nd][None], hi[nd][None])
        tn = np.maximum((pn * inv[:, :, None] - oinv[:, :, None]).max(axis=1), 0)
        tf = np.minimum((pf * inv[:, :, None] - oinv[:, :, None]).min(axis=1), tbest[:, None])
        h32 = (tn <= tf) & ~unused[nd][None]
        # f16 test as the kernel does it
        ai = np.abs(inv); flat = ~(ai <= 1024)
        S = (0.001953125 * (pmax * ai + np.abs(oinv)) + 1e-6).astype(np.float32)
        iv = np.where(flat, 0, inv).astype(np.float16)
        on = np.where(flat, 65504, oinv + S).astype(np.float16); of = np.where(flat, -65504, oinv - S).astype(np.float16)
        neg16 = np.signbit(iv)
        pn16 = np.where(neg16[:, :, None], hi16[nd][None], lo16[nd][None]); pf16 = np.where(neg16[:, :, None], lo16[nd][None], hi16[nd][None])
        with np.errstate(all='ignore'):
            tn16 = (pn16.astype(np.float32) * iv.astype(np.float32)[:, :, None] - on.astype(np.float32)[:, :, None]).astype(np.float16)
            tf16 = (pf16.astype(np.float32) * iv.astype(np.float32)[:, :, None] - of.astype(np.float32)[:, :, None]).astype(np.float16)
            tnm = np.fmax(np.fmax(tn16[:, 0], tn16[:, 1]), np.fmax(tn16[:, 2], np.float16(0)))
            tb16 = (tbest * 1.002).astype(np.float16)
            tfm = np.fmin(np.fmin(tf16[:, 0], tf16[:, 1]), np.fmin(tf16[:, 2], tb16[:, None]))
            h16 = ~np.signbit((tfm.astype(np.float32) - tnm.astype(np.float32)).astype(np.float16)) & ~unused[nd][None]
        bad += int((h32 & ~h16).sum()); extra += int((h16 & ~h32).sum()); total += int(h32.sum())
    print(name, 'nodes', nw.value, 'f32 hits', total, 'missed by f16', bad, 'extra by f16', extra, 'pmax', pmax, flush=True)
reset_all()
