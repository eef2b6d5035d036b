#!/usr/bin/env python3
'''GPU-box A/B: get_image() through a device buffer + DMA (zero_copy = 0) or written straight into the caller's
page-locked array by the resolve pass (zero_copy = 1); the benchmark step (render(32) + get_image()), alternated.'''
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from ptina_amd import scenes, common
from ptina_amd.common import ctx
from ptina_amd.things import FilmTable
from helpers import setup_engine

n = int(os.environ.get('N', 512))
eng = setup_engine(scenes.get_scene('s978'), n, n, mode='fast', max_filmsize=max(n * n, 1 << 18))
c = ctx(); c.set_option('batch', 32)
imgs = {}
for rnd in range(4):
    for z in (0, 1):
        c.set_option('zero_copy', z)
        for _ in range(3):
            eng.render(32); FilmTable().get_image()
        t0 = time.perf_counter()
        for _ in range(20):
            eng.render(32); img = FilmTable().get_image()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        imgs[z] = img.copy()
        print(f'zero_copy {z}: {dt:.4f} ms per step', flush=True)
print('same image:', np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32)) or float(np.abs(imgs[0]-imgs[1]).max()))
common.reset_all()
