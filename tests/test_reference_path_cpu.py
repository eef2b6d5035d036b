'''
The C oracle held, END TO END, to films rendered by the reference's own renderer source
(tests/golden/reference_path.npz, made by tests/golden/make_reference_path_golden.py: /root/reference's
PathEngine / path_trace / LinearBVH / GlobalStack / pools / SobolSampler executed as plain Python on
numpy scalars with the `taichi` stand-in, exams/benchmark.py's call sequence, single and double precision).

Checked per case: the LBVH arrays (Morton codes, leaf order, children: exact; boxes: exact), the Sobol
state after the reference's reset (exact), the per-pixel sample counts (exact) and the raw radiance sums
(f64 build: 1e-12 relative, measured 1e-15; f32 build: 1e-4 relative, i.e. a few ulp amplified through five bounces).
Any difference in traversal order, in which Sobol dimension feeds which decision, in the MIS weights, in
the `avoid` / light `break` / hemisphere quirks or in the film accumulation would show here as O(1).

This pins the restatement's LOGIC to the reference's source.  The arithmetic underneath is numpy's, not
Taichi's, so parity with real PTina output remains formally unpinned.
'''

import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden', 'reference_path.npz')
sys.path.insert(0, os.path.join(HERE, 'golden'))

import make_reference_path_golden as G   # noqa: E402  (scene definitions only; nothing of the reference is imported)


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLD)


@pytest.mark.parametrize('prec', ['f32', 'f64'])
@pytest.mark.parametrize('name', sorted(G.CASES))
def test_oracle_renders_what_the_reference_source_renders(gold, oracle_mod, prec, name):
    from helpers import setup_oracle
    key, nx, ny, spp = G.CASES[name]
    assert [int(x) for x in gold[f'{prec}/{name}/size']] == [nx, ny, spp]
    scene, lights, world = G.scene_of(key)
    # the world factor goes through the f32 setter like every other scene parameter (the generator's f64 run
    # uses the same f32-rounded 0.1)
    o = setup_oracle(oracle_mod, scene, nx, ny, lights=lights, world=world, f64=(prec == 'f64'), threads=2)
    n = scene[1].shape[0]

    # ---- tree/lbvh.py:169-305
    t = o.get_tree(n)
    for k in ('mc', 'leaf', 'child'):
        assert np.array_equal(t[k].astype(np.int64), gold[f'{prec}/{name}/tree/{k}']), f'LBVH {k}'
    for k in ('bmin', 'bmax'):
        # boxes are min / max of f32 vertex coordinates: exact (the f64 oracle stores them as f32 like the arrays it exports)
        assert np.array_equal(t[k].astype(np.float64), gold[f'{prec}/{name}/tree/{k}'].astype(np.float32).astype(np.float64)), f'LBVH {k}'

    # ---- sampling/sobol.py:92-105: 64 skipped points
    time_, X, P = o.sobol_state()
    assert time_ == int(gold[f'{prec}/sobol/time_after_reset']) == 64
    assert np.array_equal(np.asarray(X, np.int64), gold[f'{prec}/sobol/X_after_reset'])

    # ---- exams/benchmark.py:25-33
    o.render(1)
    o.clear()
    o.render(spp)
    film = o.get_film_real().astype(np.float64)
    want = gold[f'{prec}/{name}/film']
    assert np.array_equal(film[:, 3], want[:, 3]) and np.all(film[:, 3] == spp)
    err = np.abs(film[:, :3] - want[:, :3]) / (np.abs(want[:, :3]) + 1e-3 * spp)
    worst = float(err.max())
    print(f'{prec} {name}: worst relative difference of a pixel sum {worst:.2e}, mean radiance {want[:, :3].mean() / spp:.4f}')
    assert worst <= (1e-12 if prec == 'f64' else 1e-4), f'{prec} {name}: worst relative difference {worst:.2e}'
    assert int(gold[f'{prec}/{name}/sobol_time']) == 64 + 1 + spp + 2      # recorded after the two preview frames below
    # ---- engine/preview.py:18-41, two more frames on the same sampler: albedo -> pass 1, normal -> pass 2
    o.render_preview()
    o.render_preview()
    for pas, key in ((1, 'preview_albedo'), (2, 'preview_normal')):
        got, ref = o.get_film_real(pas).astype(np.float64), gold[f'{prec}/{name}/{key}']
        assert np.array_equal(got[:, 3], ref[:, 3]) and np.all(got[:, 3] == 2)
        e = float(np.abs(got[:, :3] - ref[:, :3]).max())
        assert e <= (1e-12 if prec == 'f64' else 2e-6), f'{prec} {name} {key}: max difference {e:.2e}'


def test_cases_exercise_the_interesting_paths(gold):
    '''the vectors are not trivial: lit pixels, dark pixels, both light types, several materials'''
    for name in G.CASES:
        f = gold[f'f64/{name}/film']
        lum = f[:, :3].sum(axis=1) / f[:, 3]
        assert lum.max() > 2.5 * lum.min() and lum.max() > 1.0, name
    scene, lights, _ = G.scene_of('lobes')
    assert {l[3] for l in lights} == {'AREA', 'POINT'}


def test_lbvh_of_the_benchmark_scene_is_the_reference_sources(gold, oracle_mod):
    '''tree/lbvh.py:169-305 run by the reference's own source on the 978-triangle scene of BASELINE configs[1]:
    Morton codes, leaf order, children and boxes equal the oracle's (tests/test_parity_gpu.py holds the
    device-built tree to the oracle's in turn)'''
    from helpers import setup_oracle
    from ptina_amd import scenes
    scene = scenes.scene_s978()
    o = setup_oracle(oracle_mod, scene, 16, 16, threads=1)
    n = scene[1].shape[0]
    t = o.get_tree(n)
    for k in ('mc', 'leaf', 'child'):
        assert np.array_equal(t[k].astype(np.int64), gold[f'f32/s978/tree/{k}']), k
    for k in ('bmin', 'bmax'):
        assert np.array_equal(t[k].astype(np.float64), gold[f'f32/s978/tree/{k}']), k
