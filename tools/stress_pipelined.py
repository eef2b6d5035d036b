#!/usr/bin/env python3
'''Diagnostic: 2048 x 2048, render(256) = eight launches back to back.  The whole film is rendered once; then the stripe shares
r = 0..7 of eight (mpt_set_stripes(16, r, 8)) are rendered in fresh contexts, `rounds` times over, and each share's columns are
compared with the whole film's, bit for bit.  usage: stress_pipelined.py [rounds] [poison] [key=value options for every context ...]'''
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
from ptina_amd import scenes, common  # noqa: E402
from ptina_amd.common import ctx  # noqa: E402
from ptina_amd.things import FilmTable  # noqa: E402
from ptina_amd.dist import stripe_columns  # noqa: E402
from helpers import setup_engine  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
poison = 'poison' in sys.argv[2:]
opts = [(a.split('=')[0], int(a.split('=')[1])) for a in sys.argv[2:] if '=' in a]
n, spp, R = 2048, 256, 8
scene = scenes.scene_s978()


def render(share):
    if poison:
        # a context whose only launch finalises its own tiles leaves its tag in every entry of a slab as large as the next
        # context's; the allocator hands that memory out again
        common.reset_all()
        eng = setup_engine(scenes.scene_s34(), n, n if share is None else n // R, mode='fast', max_filmsize=n * n)
        eng.render(32)
        FilmTable().get_raw()
    common.reset_all()
    eng = setup_engine(scene, n, n, mode='fast', max_filmsize=n * n)
    c = ctx()
    for k, v in opts:
        c.set_option(k, v)
    if share is not None:
        c.call('mpt_set_stripes', 16, share, R)
    eng.render(spp)
    return FilmTable().get_raw().reshape(n, n, 4).copy()


full = render(None)
print('whole film: counted', float(full[..., 3].min()), float(full[..., 3].max()), 'options', opts, flush=True)
bad = 0
for i in range(rounds):
    for r in range(R):
        part = render(r)
        cols = stripe_columns(n, R, r)
        d = (part[cols].view(np.uint32) != full[cols].view(np.uint32)).any(axis=-1)
        if d.any():
            bad += 1
            cx, cy = np.nonzero(d)
            diff = part[cols][cx, cy, :3] - full[cols][cx, cy, :3]
            print(f'round {i} share {r}: {len(cx)} pixels differ; share columns {cx.min()}..{cx.max()}, rows {cy.min()}..{cy.max()}; '
                  f'w {np.unique(part[cols][cx, cy, 3])}; diff range {diff.min():.4g}..{diff.max():.4g}; '
                  f'first {[(int(cols[a]), int(b)) for a, b in zip(cx[:6], cy[:6])]} diffs {diff[:4].round(4).tolist()}', flush=True)
    print(f'round {i} done, bad so far {bad}', flush=True)
common.reset_all()
print('BAD' if bad else 'OK', bad)
