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
    for key in ('lds',):                # experiment switches (tools/gpu_round.sh)
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


# The calibrated parity bounds: per-pixel L2 tolerance relative to
# (1 + |ref|), largest fraction of pixels allowed outside it, relative RMSE over the image.
#   strict build (reference order, IEEE, no contraction) vs the f32 oracle: differs by libm only
#   fast build (FMA, v_rcp / v_rsq / v_sin, ordered + culled traversal): a flipped discrete decision
#       moves a pixel by O(sample / spp), so the bound is statistical
STRICT = (1e-4, 0.001, 1e-4)
FAST = (1e-3, 0.005, 1e-3)        # rel-RMSE 1e-3: the tolerance BASELINE.md section 2 / SURVEY 8(d) state (round 3 had 2e-3; measured 2e-4 ... 6e-4)


def bounds(mode):
    return STRICT if mode == 'strict' else FAST


def _report(msg):
    '''every parity measurement of a test run also goes to gpurun_out/parity_report.txt (calibration record)'''
    import os
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'parity_report.txt'), 'a') as f:
            f.write(os.environ.get('PYTEST_CURRENT_TEST', '').split(' ')[0] + ' | ' + msg + '\n')
    except OSError:
        pass


def tile_means(img, t=8):
    '''mean rgb over t x t pixel tiles (ragged edges dropped)'''
    nx, ny = img.shape[0] // t * t, img.shape[1] // t * t
    a = img[:nx, :ny, :3].astype(np.float64)
    return a.reshape(nx // t, t, ny // t, t, 3).mean(axis=(1, 3))


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
    _report(msg)
    assert np.isfinite(img).all(), what + ': non-finite pixels'
    assert frac <= max_outlier_frac, msg
    assert rel_rmse <= rel_rmse_tol, msg
    return rel_rmse, frac


def soak_finalisation(launches, nx=512, ny=512, frames=4, per_round=50, stripes=None, stress_mb=0, stress_copies=0, log=None, opts=()):
    '''The tail finalisation's hand-off (render_kernel.hip store_sample / finalise_tiles: a sample entry of the slab is two
    8-byte granules that carry the launch's tag) checked on DATA: `per_round` launches of `frames` frames, each followed by a
    get_image() (so every launch finds the GPU idle and finalises its own tiles), replayed from the same Sobol index with the
    combine pass (option finalise = 0) once -- then round after round with the finalisation on, and the raw film must be the
    combine pass's bit for bit every time.  A sample accepted before its data had arrived (a stale or torn entry) changes a sum.
    stripes = (width, rank, world): a 1/world share as `bench.py --gpus world` deals it; stress_mb / stress_copies: that many
    device-to-device copies of that size enqueued beside every round (mpt_stress_copies); opts: context options, e.g. (('lds', 0),)
    for the gather kernels.  Returns the launches checked.'''
    from ptina_amd import scenes
    from ptina_amd.common import ctx, reset_all
    from ptina_amd.things import FilmTable
    from ptina_amd.sampling.sobol import SobolSampler
    reset_all()
    eng = setup_engine(scenes.scene_s978(), nx, ny, mode='fast', max_filmsize=max(nx * ny, 1 << 18))
    c = ctx()
    c.set_option('batch', frames)
    for k, v in opts:
        c.set_option(k, v)
    if stripes:
        c.call('mpt_set_stripes', *stripes)
    film, sob = FilmTable(), SobolSampler()

    def one_round(fin):
        c.set_option('finalise', fin)
        sob.reset()
        film.clear()
        for _ in range(per_round):
            eng.render(frames)
            film.get_image()
            assert c.get_option('last_finalised') == fin
        return film.get_raw()

    want = one_round(0).view(np.uint32).copy()
    assert np.all(want.view(np.float32)[:, 3] % (per_round * frames) == 0)
    done, rounds = 0, 0
    while done < launches:
        if stress_mb and stress_copies:
            c.call('mpt_stress_copies', int(stress_mb), int(stress_copies))
        got = one_round(1).view(np.uint32)
        if stress_mb and stress_copies:
            c.call('mpt_stress_copies', int(stress_mb), 0)
        bad = np.flatnonzero((got != want).any(axis=1))
        assert bad.size == 0, ('finalised film differs from the combine pass film in %d pixels after %d launches: first %d, got %s, want %s'
                               % (bad.size, done, bad[0], got[bad[0]].view(np.float32), want[bad[0]].view(np.float32)))
        done += per_round
        rounds += 1
        if log and rounds % 20 == 0:
            log('  %d launches bit-identical (%dx%d, %d frames per launch, stripes %s, stress %d x %d MiB per round, options %s)'
                % (done, nx, ny, frames, stripes, stress_copies, stress_mb, dict(opts)))
    reset_all()
    return done
