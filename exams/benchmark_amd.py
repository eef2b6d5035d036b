#!/usr/bin/env python3
'''
PTina's benchmark harness (reference exams/benchmark.py:8-38) driven through ptina_amd: same
objects, same call order -- init_things, PathEngine, FilmTable.set_size, pools, BVHTree.build,
Camera.set_perspective, a warm-up frame + clear, then N x PathEngine().render() and one
get_image() inside the timed region.  The glTF asset of the original is not distributed, so the
scene comes from ptina_amd.scenes (same triangle counts as the README's rows).

    python exams/benchmark_amd.py [--scene s978|s34] [--size 512] [--spp 32] [--save out.png]
'''
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ptina_amd.things import *              # noqa: E402,F401,F403
from ptina_amd.engine.path import *         # noqa: E402,F401,F403
from ptina_amd import scenes                # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--scene', default='s978')
ap.add_argument('--size', type=int, default=512)
ap.add_argument('--spp', type=int, default=32)
ap.add_argument('--save', default=None)
args = ap.parse_args()

ti.init(ti.cuda)
init_things()
PathEngine()
FilmTable().set_size(args.size, args.size)

vertices, mtlids, materials, images = scenes.get_scene(args.scene)
ModelPool().load(vertices, mtlids)
MaterialPool().load(materials)
ImagePool().load(images)
BVHTree().build()
Camera().set_perspective(scenes.BENCH_CAMERA)

PathEngine().render()
FilmTable().get_image()
FilmTable().clear()

t0 = time.time()
for i in range(args.spp):
    PathEngine().render()
img = FilmTable().get_image()
dt = time.time() - t0

title = f'{args.spp / dt:.03f} sps = {args.size * args.size * args.spp / dt / 1e6:.1f} Msamples/s'
ti.imshow(img, title)
print(title)
if args.save:
    ti.imwrite(img, args.save)
