'''shared set-up for the parity tests: the same scene driven through ptina_amd (HIP, via the
C ABI) and through the CPU oracle, with PTina's call sequence (exams/benchmark.py:8-33)'''

import numpy as np

from ptina_amd import scenes


def setup_engine(scene, nx, ny, mode='fast', camera=scenes.BENCH_CAMERA, lights=None, world=None,
                 slab=None, **caps):
    from ptina_amd.things import init_things, FilmTable, ModelPool, MaterialPool, ImagePool, \
        BVHTree, Camera, LightPool, WorldLight
    from ptina_amd.engine.path import PathEngine
    from ptina_amd.common import ctx
    from ptina_amd import _lib
    init_things(**caps)
    eng = PathEngine()
    ctx().set_option('mode', _lib.MODE_STRICT if mode == 'strict' else _lib.MODE_FAST)
    import os
    for key in ('sched_num', 'sched_den', 'lds'):                # experiment switches (tools/gpu_round.sh)
        if os.environ.get('MIPTINA_' + key.upper()):
            ctx().set_option(key, int(os.environ['MIPTINA_' + key.upper()]))
    FilmTable().set_size(nx, ny)
    vertices, mtlids, materials, images = scene
    ModelPool().load(vertices, mtlids)
    MaterialPool().load(materials)
    ImagePool().load(images)
    BVHTree().build()
    Camera().set_perspective(camera)
    if lights is not None:
        LightPool().clear()
        for l in lights:
            LightPool().add(*l)
    if world is not None:
        WorldLight().set(*world)
    if slab is not None:
        ctx().call('mpt_set_slab', int(slab[0]), int(slab[1]))
    return eng


def setup_oracle(oracle_mod, scene, nx, ny, camera=scenes.BENCH_CAMERA, lights=None, world=None,
                 f64=False, threads=None):
    o = oracle_mod.Oracle(f64=f64, threads=threads)
    o.set_size(nx, ny)
    o.load_scene(scene, camera)
    if lights is not None:
        o.clear_lights()
        for l in lights:
            o.add_light(*l)
    if world is not None:
        o.set_world_light(*world)
    return o


def image_stats(img, ref):
    '''per-pixel L2 over rgb, relative RMSE, fraction of pixels outside tol'''
    a = img[..., :3].astype(np.float64)
    b = ref[..., :3].astype(np.float64)
    d = np.sqrt(((a - b) ** 2).sum(axis=-1))
    refn = np.sqrt((b ** 2).sum(axis=-1))
    rel_rmse = float(np.sqrt((d ** 2).mean()) / max(np.sqrt((refn ** 2).mean()), 1e-30))
    return d, refn, rel_rmse


def assert_parity(img, ref, pix_tol, max_outlier_frac, rel_rmse_tol, what=''):
    d, refn, rel_rmse = image_stats(img, ref)
    bad = d > pix_tol * (1.0 + refn)
    frac = float(bad.mean())
    msg = (f'{what}: rel-RMSE {rel_rmse:.3e} (tol {rel_rmse_tol:.1e}), outliers {frac:.4%} '
           f'(tol {max_outlier_frac:.2%}) at per-pixel L2 tol {pix_tol:.1e}*(1+|ref|), max diff {d.max():.3e}')
    print(msg)
    assert np.isfinite(img).all(), what + ': non-finite pixels'
    assert frac <= max_outlier_frac, msg
    assert rel_rmse <= rel_rmse_tol, msg
    return rel_rmse, frac
