import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from ptina_amd import scenes
from ptina_amd.common import ctx, reset_all
from ptina_amd.things import FilmTable
from helpers import setup_engine, tile_means
LOBE = {'glass': dict(basecolor=(0.9, 0.95, 1.0), roughness=0.08, transmission=0.9, ior=1.5, specular=0.5),
        'rough_glass': dict(basecolor=(0.8, 0.9, 0.8), roughness=0.45, transmission=0.6, ior=1.33, metallic=0.1)}
def scene(name, lift):
    parts = [scenes.cornell_walls(), scenes.box((-0.7, 1.2 + lift, -0.6), (0.6, 1.2, 0.6), 18.0, 3), scenes.box((0.75, 0.6 + lift, 0.55), (0.6, 0.6, 0.6), -17.0, 4)]
    v, m = scenes._compose(parts)
    mats = list(scenes.WALL_MATERIALS) + [scenes.material(**LOBE[name]), scenes.material(**LOBE[name])]
    return v, m, mats, []
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
for name in ('glass', 'rough_glass'):
    for lift in (0.0, 0.01):
        imgs = {}
        for mode, opts in (('strict', ()), ('fast', ()), ('fast_nolds', (('lds', 0),)), ('fast_bin', (('lds_wide', 0),))):
            reset_all()
            eng = setup_engine(scene(name, lift), 32, 32, mode=mode.split('_')[0])
            for k, v in opts: ctx().set_option(k, v)
            eng.render(spp)
            imgs[mode] = FilmTable().get_image().copy()
        reset_all()
        sc = float(imgs['strict'][..., :3].mean())
        for mode in ('fast', 'fast_nolds', 'fast_bin'):
            d = (tile_means(imgs[mode]) - tile_means(imgs['strict'])) / sc
            print(name, 'lift', lift, mode, 'mean diff %.4f%%' % (100 * (imgs[mode][..., :3].mean() / sc - 1)), 'tile rms %.4f%%' % (100 * np.sqrt((d ** 2).sum(-1).mean())))
            if mode == 'fast': print(np.round(100 * d.sum(-1) / 3, 3))
# how often the near-tie path runs (counting build), coincident scene
reset_all()
eng = setup_engine(scene('glass', 0.0), 32, 32, mode='fast')
ctx().set_option('count', 1); ctx().call('mpt_reset_counters')
eng.render(64); ctx().call('mpt_flush')
k = ctx().counters()
print('near-ties decided by the reference rule: %d of %d triangle tests, %d rays' % (k['pl_taken'], k['n_tri'], k['rays']))
reset_all()
