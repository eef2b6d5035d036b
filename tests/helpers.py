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


def stress_scene(seed):
    '''the seeded random scene of tools/stress_random_scenes.py / test_forty_random_scenes_*: the cornell walls plus 1 .. 900
    triangles of random place, size and (smooth) normals, 2 .. 5 random opaque Disney materials, 1 .. 3 point / area lights, a
    random constant world light, a random film size, sample count and batch size.  Returns (scene, lights, world, nx, ny, spp,
    batches) -- batches: one batch size per kernel run, drawn in the tool's order'''
    from ptina_amd.tools.matrix import translate
    rng = np.random.default_rng(seed)
    walls = scenes.cornell_walls()
    k = int(rng.integers(1, 900))
    c = rng.uniform([-1.6, 0.3, -1.6], [1.6, 3.4, 1.2], (k, 1, 3))
    P = c + rng.normal(0, 1, (k, 3, 3)) * rng.uniform(0.03, 0.7, (k, 1, 1))
    fn = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    fn /= np.linalg.norm(fn, axis=1, keepdims=True) + 1e-30
    N = fn[:, None, :] + rng.normal(0, 0.25, (k, 3, 3))
    N /= np.linalg.norm(N, axis=2, keepdims=True)
    T = rng.uniform(0, 1, (k, 3, 2))
    nm = int(rng.integers(2, 6))
    M = rng.integers(3, 3 + nm, k).astype(np.int32)
    v, m = scenes._compose([walls, (P, N, T, M)])
    mats = list(scenes.WALL_MATERIALS)
    for _ in range(nm):
        mats.append(scenes.material(basecolor=tuple(rng.uniform(0, 1, 3) * (rng.random() > 0.15)), metallic=float(rng.random() ** 2),
                                    roughness=float(rng.uniform(0.05, 1)), specular=float(rng.random()), specularTint=float(rng.random()),
                                    subsurface=float(rng.random() * (rng.random() > 0.5)), sheen=float(rng.random() * (rng.random() > 0.5)),
                                    sheenTint=float(rng.random())))
    scene = (v, m, mats, [])
    rot = np.eye(4)
    rot[:3, :3] = [[1, 0, 0], [0, 0, 1], [0, -1, 0]]
    lights = []
    for _ in range(int(rng.integers(1, 4))):
        pos = rng.uniform([-1.5, 2.2, -1.5], [1.5, 3.8, 1.5])
        if rng.random() < 0.5:
            lights.append((translate(list(pos)) @ rot, rng.uniform(4, 20, 3), float(rng.uniform(0.2, 0.7)), 'AREA'))
        else:
            lights.append((translate(list(pos)), rng.uniform(4, 20, 3), float(rng.uniform(0.05, 0.4)), 'POINT'))
    world = ([float(x) for x in rng.uniform(0, 0.4, 3)] + [1.0], -1)
    nx, ny, spp = int(rng.integers(20, 200)), int(rng.integers(20, 160)), int(rng.integers(1, 40))
    batches = [int(rng.integers(1, 33)) for _ in range(5)]
    return scene, lights, world, nx, ny, spp, batches


STRESS_KERNELS = (('strict', 'strict', {}), ('lds4', 'fast', {}), ('lds', 'fast', {'lds_wide': 0}), ('bin', 'fast', {'lds': 0, 'wide': 0}),
                  ('wide', 'fast', {'lds': 0}))


def _largest_sample(scene, nx, ny, spp, mode, lights, world):
    '''per pixel: the largest single-sample radiance (L2 over rgb) of the film rendered one frame at a time -- the same frames,
    the same sums (batching does not change a film), read back after every frame'''
    from ptina_amd.common import ctx, reset_all
    from ptina_amd.things import FilmTable
    reset_all()
    eng = setup_engine(scene, nx, ny, mode=mode, lights=lights, world=world)
    ctx().set_option('batch', 1)
    prev = np.zeros((nx, ny, 3), np.float64)
    big = np.zeros((nx, ny), np.float64)
    for _ in range(spp):
        eng.render(1)
        raw = FilmTable().get_raw().reshape(nx, ny, 4)[..., :3].astype(np.float64)
        big = np.maximum(big, np.sqrt(((raw - prev) ** 2).sum(axis=-1)))
        prev = raw
    return big


def stress_compare(seed, log=print):
    '''One random scene through the strict build and the four production kernels, each production film against the strict film at
    the STATED fast bounds (helpers.FAST: 99.5 % of the pixels within 1e-3 (1 + |ref|), rel-RMSE <= 1e-3) with explicit
    firefly accounting: a path whose discrete decision (lobe, edge hit, shadow) falls the other way replaces
    ONE sample of a pixel, so such a pixel may leave the per-pixel bound by at most the largest single sample of that pixel in
    either film / spp; at most ceil(0.0005 x pixels) pixels may claim that, they still count as outliers, and the rel-RMSE is
    taken without them.  A pixel that is further off than one sample explains fails the scene.
    Returns (ok, fireflies, messages).'''
    import math
    from ptina_amd.common import ctx, reset_all
    from ptina_amd.things import FilmTable
    scene, lights, world, nx, ny, spp, batches = stress_scene(seed)
    imgs, kernels = {}, {}
    for (name, mode, opts), batch in zip(STRESS_KERNELS, batches):
        reset_all()
        eng = setup_engine(scene, nx, ny, mode=mode, lights=lights, world=world)
        for a, b in opts.items():
            ctx().set_option(a, b)
        ctx().set_option('batch', batch)
        eng.render(spp)
        raw = FilmTable().get_raw().reshape(nx, ny, 4)
        assert np.all(raw[..., 3] == spp), (seed, name)
        imgs[name] = FilmTable().get_image()
        kernels[name] = ctx().get_option('last_kernel')
    ok, msgs, flies, big = True, [], 0, None
    kmax = math.ceil(0.0005 * nx * ny)
    for name in ('lds4', 'lds', 'bin', 'wide'):
        d, refn, rel = image_stats(imgs[name], imgs['strict'])
        tol = FAST[0] * (1 + refn)
        out = d > tol
        k_used, rel_used = 0, rel
        if rel > FAST[2] or out.mean() > FAST[1]:
            # only now the (expensive) per-sample films: which of the outliers can one replaced sample explain?
            if big is None:
                big = np.maximum(_largest_sample(scene, nx, ny, spp, 'strict', lights, world),
                                 _largest_sample(scene, nx, ny, spp, 'fast', lights, world))
            fly = out & (d <= tol + big / spp * (1 + 1e-3) + 1e-6)
            order = np.argsort(-(d * fly).ravel())[:kmax]                # the kmax worst explainable pixels
            mask = np.zeros(nx * ny, bool)
            mask[order] = True
            mask &= fly.ravel()
            k_used = int(mask.sum())
            d2 = d.copy().ravel()
            d2[mask] = 0.0
            rel_used = float(np.sqrt((d2 ** 2).mean()) / max(np.sqrt((refn ** 2).mean()), 1e-30))
            unexplained = out & ~fly
            if unexplained.any():
                ix = np.argwhere(unexplained)[0]
                msgs.append(f'{name}: pixel {tuple(ix)} is {d[tuple(ix)]:.3e} off with tolerance {tol[tuple(ix)]:.3e} and a largest sample / spp of '
                            f'{big[tuple(ix)] / spp:.3e}: more than one replaced sample explains')
                ok = False
        good = rel_used <= FAST[2] and out.mean() <= FAST[1]
        ok = ok and good
        flies = max(flies, k_used)
        msgs.append(f'{name} rel {rel:.1e}' + (f' ({rel_used:.1e} without {k_used} firefly pixel(s) of at most {kmax})' if k_used else '') +
                    f' out {out.mean() * 100:.2f}%' + ('' if good else ' <<<<'))
    same = np.array_equal(imgs['lds'].view(np.uint32), imgs['bin'].view(np.uint32))
    same4 = float((imgs['lds4'].view(np.uint32) == imgs['lds'].view(np.uint32)).all(axis=-1).mean())
    log(f'{seed} {scene[1].shape[0]} tris {nx}x{ny} {spp} spp | ' + ' | '.join(msgs) + f' | lds==bin {same} | lds4==lds on {100 * same4:.3f} % of the pixels | '
        f'fireflies {flies} of at most {kmax} | kernels {kernels}')
    reset_all()
    return ok and same, flies, msgs
