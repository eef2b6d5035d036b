'''
GPU parity tests (-m gpu): the HIP path, driven through the C ABI by the PTina-named Python
classes, against the CPU oracle on the same seeded inputs.

Tolerances (resolved image, per-pixel L2 over rgb; helpers.STRICT / helpers.FAST), calibrated on
MI355X and against the oracle's own f32-vs-f64 spread
(test_oracle_cpu.py::test_f32_oracle_agrees_with_f64_build):
  strict build : >= 99.9 % of pixels within 1e-4 * (1 + |ref|), relative RMSE <= 1e-4
  fast build   : >= 99.5 % of pixels within 1e-3 * (1 + |ref|), relative RMSE <= 1e-3;
                 one flipped discrete decision (lobe choice, edge hit) moves a pixel by O(sample/spp),
                 so cases with few samples per pixel or ill-conditioned materials state their own
                 calibrated bound, with the measurement it comes from.
Integer / index work (Sobol state, tree arrays, sample counts) is bit-exact.
'''

import os

import numpy as np
import pytest

from ptina_amd import scenes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
from helpers import STRICT, FAST, bounds   # noqa: E402  (the calibrated parity bounds)


def _engine(fresh, *a, **kw):
    from helpers import setup_engine
    return setup_engine(*a, **kw)


def test_sobol_state_on_device_is_bit_exact(fresh):
    from ptina_amd.things import init_things
    from ptina_amd.sampling.sobol import SobolSampler
    g = np.load(os.path.join(GOLD, 'sobol_points.npz'))
    want = {int(k): p for k, p in zip(g['k'], g['P'])}
    init_things()
    s = SobolSampler()
    t, X, P = s.state()
    assert t == 64 and np.array_equal(P, want[64])
    s.update()
    t, X, P = s.state()
    assert t == 65 and np.array_equal(P, want[65])
    s.skip = 0
    s.reset()
    for k in (1, 2, 3):
        s.update()
        assert np.array_equal(s.state()[2], want[k])


@pytest.mark.parametrize('name', ['s34', 's978'])
def test_tree_is_node_for_node_the_reference_lbvh(fresh, oracle_mod, name):
    from ptina_amd.things import BVHTree
    scene = scenes.get_scene(name)
    n = scene[1].shape[0]
    _engine(fresh, scene, 16, 16)
    t = BVHTree().to_numpy()
    o = oracle_mod.Oracle(sobol=False)
    o.load_model(scene[0], scene[1])
    o.build_tree()
    r = o.get_tree(n)
    for k in ('mc', 'leaf', 'child', 'bmin', 'bmax'):
        assert np.array_equal(t[k], r[k]), k
    assert t['depth'] + 1 <= 32


def _bench_sequence(eng, film, spp):
    eng.render()                     # exams/benchmark.py:25-27
    film.get_image()
    film.clear()
    for _ in range(spp):             # :31-33, one call per sample like the reference
        eng.render()
    return film.get_image()


@pytest.mark.parametrize('name,nx,ny,spp', [('s34', 64, 64, 8), ('s978', 96, 80, 8)])
def test_strict_build_matches_oracle(fresh, oracle_mod, name, nx, ny, spp):
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import FilmTable
    scene = scenes.get_scene(name)
    ref = setup_oracle(oracle_mod, scene, nx, ny)
    ref.render(1)
    ref.clear()
    ref.render(spp)
    want = ref.get_image()
    eng = _engine(fresh, scene, nx, ny, mode='strict')
    img = _bench_sequence(eng, FilmTable(), spp)
    assert np.all(img[..., 3] == 1.0)
    assert_parity(img, want, *STRICT, what=f'strict {name} {nx}x{ny}x{spp}')


@pytest.mark.parametrize('name,nx,ny,spp', [('s34', 96, 96, 32), ('s978', 128, 128, 32)])
def test_fast_build_matches_oracle(fresh, oracle_mod, name, nx, ny, spp):
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import FilmTable
    scene = scenes.get_scene(name)
    ref = setup_oracle(oracle_mod, scene, nx, ny)
    ref.render(1)
    ref.clear()
    ref.render(spp)
    want = ref.get_image()
    eng = _engine(fresh, scene, nx, ny, mode='fast')
    img = _bench_sequence(eng, FilmTable(), spp)
    assert np.all(img[..., 3] == 1.0)
    assert_parity(img, want, *FAST, what=f'fast {name} {nx}x{ny}x{spp}')


def test_golden_fixture_film(fresh):
    '''the committed oracle film (tests/golden/oracle_s34_24x24x4.npz) vs the strict build'''
    from helpers import assert_parity
    from ptina_amd.things import FilmTable
    g = np.load(os.path.join(GOLD, 'oracle_s34_24x24x4.npz'))
    eng = _engine(fresh, scenes.scene_s34(), 24, 24, mode='strict')
    img = _bench_sequence(eng, FilmTable(), 4)
    raw = FilmTable().get_raw()
    assert np.all(raw[:, 3] == 4.0)
    assert_parity(img, g['image'], *STRICT, what='golden s34 24x24x4')


def test_film_api_semantics(fresh):
    from ptina_amd.things import FilmTable
    eng = _engine(fresh, scenes.scene_s34(), 20, 12)
    film = FilmTable()
    assert (film.nx, film.ny) == (20, 12)
    img = film.get_image()
    assert img.shape == (20, 12, 4) and np.allclose(img, [0.9, 0.4, 0.9, 0.0])   # filmtable.py:60-61
    eng.render()
    eng.render()
    raw = film.get_raw().reshape(20, 12, 4)
    img = film.get_image()
    assert np.all(raw[..., 3] == 2.0)
    assert np.array_equal(img[..., :3], raw[..., :3] / raw[..., 3:4]) and np.all(img[..., 3] == 1)
    # the same image whichever way it travels: written by the resolve pass straight into the page-locked array (default),
    # through a device buffer and a DMA (zero_copy = 0), or into an ordinary numpy array through the staging buffer
    from ptina_amd.common import ctx
    from ptina_amd._lib import fptr
    ctx().set_option('zero_copy', 0)
    assert np.array_equal(film.get_image().view(np.uint32), img.view(np.uint32))
    ctx().set_option('zero_copy', 1)
    plain = np.empty((20, 12, 4), np.float32)
    ctx().call('mpt_get_image', 0, fptr(plain))
    assert np.array_equal(plain.view(np.uint32), img.view(np.uint32))
    flat = np.zeros(20 * 12 * 3, np.float32)
    film.fast_export_image(flat)
    assert np.array_equal(flat.reshape(12, 20, 3), np.swapaxes(img[..., :3], 0, 1))   # (y*nx + x)*3
    film.clear(1)                             # clears every pass, filmtable.py:44-45
    assert np.all(film.get_raw(0) == 0)


@pytest.mark.parametrize('mode', ['strict', 'fast'])
def test_batching_does_not_change_the_film(fresh, mode):
    '''32 x render() fused into one launch == 32 single-frame launches (strict: bit for bit;
    fast: bit for bit at equal chunking)'''
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    films = []
    for batch in (1, 16):
        reset_all()
        eng = _engine(None, scenes.scene_s34(), 40, 24, mode=mode)
        ctx().set_option('batch', batch)
        ctx().set_option('chunk', 1)
        for _ in range(16):
            eng.render()
        films.append(FilmTable().get_raw())
    assert np.array_equal(films[0], films[1])
    assert np.all(films[0][:, 3] == 16)


def test_work_item_shape_does_not_change_the_film(fresh):
    '''work items are (tile, frames) pieces of the launch; a wave prepares the primary rays of 64 consecutive
    samples of its item at a time: items of 16, 32, 64, 128 and 192 samples (the last one of a launch shorter),
    on a film whose edges cut tiles, all give the same film bit for bit -- LDS-resident and gather kernels'''
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    nx, ny = 52, 43
    films = {}
    for lds in (1, 0):
        for tw, th, chunk in ((3, 3, 1), (3, 3, 2), (3, 3, 3), (3, 2, 1), (2, 2, 1), (2, 2, 7), (1, 0, 1)):
            reset_all()
            eng = _engine(None, scenes.scene_s978(), nx, ny, mode='fast')
            c = ctx()
            c.set_option('lds', lds)
            c.set_option('batch', 8)
            c.set_option('tile_w_shift', tw)
            c.set_option('tile_h_shift', th)
            c.set_option('chunk', chunk)
            eng.render(8)
            eng.render(3)
            c.call('mpt_flush')
            films[(lds, tw, th, chunk)] = FilmTable().get_raw().copy()
    reset_all()
    ref = films[(1, 3, 3, 1)]
    assert np.all(ref.reshape(nx, ny, 4)[..., 3] == 11)
    for key, film in films.items():
        assert np.array_equal(film, ref), key


def test_tail_finalisation_gives_the_combine_pass_film_and_image(fresh):
    '''option "finalise" (default 1): a render launch that has the chip to itself adds its frames to the film, resolves and writes
    out finished tiles itself while its last paths drain (render_kernel.hip finalise_tiles: write-through slab entries of two
    8-byte granules that carry the launch's tag, sc1 re-reads until both tags are there); 0 = the combine pass after the launch.  Same raw film and same
    get_image() bits, every time: for the three production kernels (LDS-resident, 4-wide gather, binary gather), ragged films,
    several tile shapes and frames per work item, several batches in a row, a film that already holds sums, column slabs
    and stripes, the counting build, and the whole 512 x 512 x 32 benchmark film five times over (a hand-off that loses a
    sample once in a million would show there)'''
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all

    def run(fin, nx, ny, frames, opts=(), slab=None, stripes=None, scene=None, count=0):
        reset_all()
        eng = _engine(None, scene or scenes.scene_s978(), nx, ny, mode='fast', slab=slab, max_filmsize=max(nx * ny, 1 << 18))
        c = ctx()
        c.set_option('finalise', fin)
        c.set_option('batch', 8)
        c.set_option('count', count)
        for k, v in opts:
            c.set_option(k, v)
        if stripes:
            c.call('mpt_set_stripes', *stripes)
        flags = []
        imgs = []
        for f in frames:
            eng.render(f)
            imgs.append(FilmTable().get_image().copy())          # (a read-back after every batch: each launch finds the ring idle)
            flags.append(c.get_option('last_finalised'))
        return FilmTable().get_raw().copy(), imgs, flags, c.get_option('last_kernel')

    cases = [dict(nx=52, ny=43, frames=(8, 3, 8)), dict(nx=52, ny=43, frames=(5,), opts=(('lds', 0),)),
             dict(nx=52, ny=43, frames=(8, 2), opts=(('lds', 0), ('wide', 0))),
             dict(nx=52, ny=43, frames=(8, 3), opts=(('tile_w_shift', 3), ('tile_h_shift', 2))),
             dict(nx=52, ny=43, frames=(8,), opts=(('tile_w_shift', 2), ('tile_h_shift', 2), ('chunk', 3))),
             dict(nx=52, ny=43, frames=(4,), opts=(('tile_w_shift', 1), ('tile_h_shift', 0))),       # two pixels per tile: 62 lanes of a finishing wave idle
             dict(nx=64, ny=40, frames=(8, 8), slab=(16, 40)), dict(nx=96, ny=40, frames=(8, 1), stripes=(16, 1, 3)),
             dict(nx=40, ny=24, frames=(4,), scene=scenes.scene_s34(), count=1)]
    for case in cases:
        ref = run(0, **case)
        got = run(1, **case)
        assert all(f == 0 for f in ref[2]) and all(f == 1 for f in got[2]), (case, ref[2], got[2])
        assert ref[3] == got[3]
        assert np.array_equal(got[0].view(np.uint32), ref[0].view(np.uint32)), case
        for a, b in zip(got[1], ref[1]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), case
    # the benchmark film, repeatedly, and pipelined launches in between (render(64) = two launches back to back: the second finds
    # the first in flight and keeps the combine pass; both orders of addition are the frame order)
    films = {}
    for fin in (0, 1):
        reset_all()
        eng = _engine(None, scenes.scene_s978(), 512, 512, mode='fast')
        c = ctx()
        c.set_option('finalise', fin)
        c.set_option('batch', 32)
        out = []
        for rep in range(5):
            eng.render(32)
            out.append(FilmTable().get_image().copy())
            assert c.get_option('last_finalised') == fin
        eng.render(64)
        out.append(FilmTable().get_image().copy())
        assert c.get_option('last_finalised') == 0          # (the second launch of the pair)
        out.append(FilmTable().get_raw().copy())
        films[fin] = out
    reset_all()
    for a, b in zip(films[1], films[0]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.all(films[1][-1][:, 3] == 5 * 32 + 64)


def test_tail_finalisation_soak_compares_the_data(fresh):
    '''VERDICT r04 next #1.  300 + 200 + 200 finalising launches, each round of 50 replayed from the same Sobol index, and the raw film
    must be the combine pass's bit for bit after every round -- the whole 512 x 512 film, a 1/8 share dealt in stripes (most of a
    finishing wave's tiles are then somebody else's), a ragged film; the second and third with a stream of 1 GiB device-to-device
    copies beside the render.  A sample accepted before its bytes had arrived would change a sum (tools/soak.py runs the same for
    20 000 launches: profiles/r05_soak.log).  Then the tags coming round: the launch numbered 65534 zeroes every slab behind the
    pending combine pass and starts the tags again, and nothing changes'''
    from helpers import soak_finalisation
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    assert soak_finalisation(300) == 300
    assert soak_finalisation(200, stripes=(16, 5, 8), stress_mb=1024, stress_copies=24) == 200
    assert soak_finalisation(200, nx=500, ny=310, frames=3, stress_mb=1024, stress_copies=24) == 200
    films = {}
    for start in (0, 65534 - 6):
        reset_all()
        eng = _engine(None, scenes.scene_s978(), 96, 80, mode='fast')
        c = ctx()
        c.set_option('batch', 4)
        c.set_option('launch_seq', start)
        for k in range(12):
            eng.render(4)
            if k % 3 == 2:
                eng.render(4)                               # a pipelined pair now and then: its second launch keeps the combine pass
            FilmTable().get_image()
        films[start] = FilmTable().get_raw().copy()
        # (the launch_seq door itself makes the next finalising launch zero the slabs: one zeroing for it, one for the tags coming round)
        assert c.get_option('tag_wraps') == (2 if start else 1)
    reset_all()
    assert np.array_equal(films[0].view(np.uint32), films[65534 - 6].view(np.uint32))
    assert np.all(films[0][:, 3] == 16 * 4)


def test_tags_come_round_at_the_head_of_a_pipelined_pair(fresh):
    '''ADVICE r05 (medium).  When the tags come round every sample slab is zeroed; the launch that finds this out is the first of a
    pipelined pair here, on a 1024 x 1024 film (32-frame slabs of 512 MiB: the memsets take a while), and the second launch of the
    pair runs on another stream and writes another slab -- which nothing ordered behind that slab's memset before the host waited
    for the memsets: the combine pass then added zeros.  Same film as the same calls far from the wrap, bit for bit, every sample
    counted'''
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    films = {}
    for start in (1000, 65534 - 2):
        reset_all()
        eng = _engine(None, scenes.scene_s978(), 1024, 1024, mode='fast', max_filmsize=1 << 20)
        c = ctx()
        c.set_option('batch', 32)
        eng.render(64)                                      # both slabs exist and hold old entries
        FilmTable().get_image()
        FilmTable().clear()
        c.set_option('launch_seq', start)
        eng.render(32)                                      # (the door's own zeroing happens here)
        FilmTable().get_image()
        w0 = c.get_option('tag_wraps')
        for _ in range(3):
            eng.render(64)                                  # start = 65532: the first launch of the first pair is number 65534
            assert c.get_option('last_finalised') == 0      # (the second launch of a pair keeps the combine pass)
        FilmTable().get_image()
        assert c.get_option('tag_wraps') - w0 == (1 if start > 60000 else 0)
        films[start] = FilmTable().get_raw().copy()
    reset_all()
    assert np.all(films[1000][:, 3] == 32 + 3 * 64)
    assert np.array_equal(films[1000].view(np.uint32), films[65534 - 2].view(np.uint32))


def test_image_hint_is_advice_only(fresh):
    '''mpt_hint_image says where the next get_image(0) will want its array, so that a finalising launch can write the resolved
    image while it drains (FilmTable does this for every PathEngine.render()).  It is advice: the image is right when the hint
    is used, when the film changes between the render and the read-back (clear, another batch, a resize), when get_image is
    given another array, when there is no hint at all, and when the hinted array is dropped unused'''
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx
    from ptina_amd._lib import fptr, host_array
    nx, ny = 72, 40
    eng = _engine(fresh, scenes.scene_s978(), nx, ny, mode='fast')
    c = ctx()
    film = FilmTable()
    c.set_option('batch', 4)

    def want():
        raw = film.get_raw().reshape(nx, ny, 4)
        img = np.empty_like(raw)
        img[..., :3] = raw[..., :3] / raw[..., 3:4]
        img[..., 3] = 1.0
        return img

    eng.render(4)                                    # hint -> launch writes the image -> get_image only waits
    assert c.get_option('last_finalised') == 1
    a = film.get_image()
    assert np.array_equal(a, want())
    eng.render(4)
    b = film.get_image()
    assert b is not a and np.array_equal(b, want()) and not np.array_equal(a, b)     # a fresh array every time; the old one untouched
    a_copy = a.copy()
    eng.render(4)
    eng.render(4)                                    # a second batch after the one that wrote the image: the early image is stale
    d = film.get_image()
    assert np.array_equal(d, want()) and np.array_equal(a, a_copy)
    eng.render(4)
    film.clear()                                     # the film changed under the early image
    e = film.get_image()
    assert np.allclose(e, [0.9, 0.4, 0.9, 0.0])
    eng.render(4)
    other = host_array((nx, ny, 4))                  # get_image into another array than the hinted one
    c.call('mpt_get_image', 0, fptr(other))
    assert np.array_equal(other, want())
    f = film.get_image()                             # (no launch since: the resolve pass)
    assert np.array_equal(f, other)
    eng.render(4)
    film._next = None                                # the hinted array dropped unused: FilmTable makes another one ...
    c.call('mpt_hint_image', 0, None)                # ... after telling the library (which waits for the launch writing into it)
    g = film.get_image()
    assert np.array_equal(g, want())
    eng.render(2)
    film.set_size(40, 24)                            # resize between render and read-back: the hinted array has the wrong shape
    film.clear()
    eng.render(4)
    h = film.get_image()
    raw = film.get_raw().reshape(40, 24, 4)
    assert h.shape == (40, 24, 4) and np.array_equal(h[..., :3], raw[..., :3] / raw[..., 3:4])


def test_workgroup_size_does_not_change_the_film(fresh):
    '''the LDS-resident kernel over the binary nodes picks its persistent workgroup by the launch's size (768 lanes below six
    samples per lane, 1024 above: miptina.cpp), the one over the 4-wide nodes always takes 1024; whatever is picked or forced -- 256, 512, 768, 1024 lanes, i.e. 1 to 4 waves per SIMD sharing
    the scheduler's passes differently -- the film is the same bit for bit, on a film small enough for the rule's small side
    and on one on its large side'''
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    for nx, ny, spp in ((52, 43, 8), (640, 512, 6)):          # 0.07 and 7.5 samples per lane of a 256-CU launch
        for lds_wide in (1, 0):                                # over the 4-wide nodes (the default) and over the binary ones
            films = {}
            for block in (0, 256, 512, 768, 1024):
                reset_all()
                eng = _engine(None, scenes.scene_s978(), nx, ny, mode='fast', max_filmsize=max(nx * ny, 1 << 18))
                c = ctx()
                c.set_option('batch', 8)
                c.set_option('lds_block', block)
                c.set_option('lds_wide', lds_wide)
                eng.render(spp)
                c.call('mpt_flush')
                assert c.get_option('last_kernel') == (5 if lds_wide else 1)
                films[block] = FilmTable().get_raw().copy()
            reset_all()
            assert np.all(films[0].reshape(nx, ny, 4)[..., 3] == spp)
            for block, film in films.items():
                assert np.array_equal(film.view(np.uint32), films[0].view(np.uint32)), (nx, ny, lds_wide, block)


def test_slabs_reassemble_bit_identically(fresh):
    '''two column slabs rendered separately == the full film (what the multi-GPU path relies on)'''
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    nx, ny, spp = 50, 37, 4                   # ragged: not multiples of the 16x16 tile
    eng = _engine(None, scenes.scene_s34(), nx, ny, mode='fast')
    eng.render(spp)
    full = FilmTable().get_raw().reshape(nx, ny, 4)
    parts = np.zeros_like(full)
    for x0, x1 in ((0, 23), (23, 50)):
        reset_all()
        eng = _engine(None, scenes.scene_s34(), nx, ny, mode='fast', slab=(x0, x1))
        eng.render(spp)
        got = FilmTable().get_raw().reshape(nx, ny, 4)
        assert np.all(got[:x0] == 0) and np.all(got[x1:] == 0)
        parts[x0:x1] = got[x0:x1]
    reset_all()
    assert np.array_equal(parts, full)
    assert np.all(full[..., 3] == spp)


def test_stripes_reassemble_bit_identically(fresh):
    '''the film dealt out in 16-column stripes to three contexts == the full film, fast and strict
    build and the preview passes (the load-balanced multi-GPU split)'''
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    from ptina_amd.dist import stripe_columns
    from ptina_amd.engine.preview import PreviewEngine
    nx, ny, spp, world = 102, 37, 3, 3        # 6 full stripes + one of 6 columns
    for mode in ('fast', 'strict'):
        reset_all()
        eng = _engine(None, scenes.scene_s34(), nx, ny, mode=mode)
        eng.render(spp)
        PreviewEngine().render()
        full = [FilmTable().get_raw(p).reshape(nx, ny, 4).copy() for p in range(3)]
        parts = [np.zeros_like(f) for f in full]
        for r in range(world):
            reset_all()
            eng = _engine(None, scenes.scene_s34(), nx, ny, mode=mode)
            ctx().call('mpt_set_stripes', 16, r, world)
            eng.render(spp)
            PreviewEngine().render()
            cols = stripe_columns(nx, world, r)
            other = np.setdiff1d(np.arange(nx), cols)
            for p in range(3):
                got = FilmTable().get_raw(p).reshape(nx, ny, 4)
                assert np.all(got[other] == 0), (mode, r, p)
                parts[p][cols] = got[cols]
        for p in range(3):
            assert np.array_equal(parts[p], full[p]), (mode, p)
        assert np.all(full[0][..., 3] == spp)
    reset_all()


def test_long_interactive_session(fresh):
    '''a viewport-style session: thousands of one-frame launches with a read-back now and then (the
    launch-timing bookkeeping must stay bounded); the film ends up with every frame, bit-identical to
    the same frames rendered in batches'''
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    nx, ny, frames = 24, 16, 4500
    eng = _engine(None, scenes.scene_s34(), nx, ny, mode='fast')
    ctx().set_option('batch', 1)
    for f in range(frames):
        eng.render()
        if f % 1000 == 999:
            assert np.all(FilmTable().get_raw()[:, 3] == f + 1)
    one = FilmTable().get_raw().copy()
    ms, launches = ctx().kernel_time()
    assert 0 < launches <= 4096 and ms > 0
    reset_all()
    eng = _engine(None, scenes.scene_s34(), nx, ny, mode='fast')
    eng.render(frames)
    assert np.array_equal(FilmTable().get_raw(), one)
    reset_all()


def test_lds_and_gather_kernels_agree_bit_for_bit(fresh):
    '''the LDS-resident kernel and the gather kernel over the binary tree run the same state machine on the
    same tree: one film, whichever serves the scene; the SAH and the plain-LBVH tree, and the gather kernel over
    the 4-wide collapse of either -- with exact child boxes or with the production records that hold them in 8 bits,
    rounded outwards -- visit the same triangles in another order: equal up to equal-depth ties'''
    from helpers import assert_parity
    from ptina_amd.things import FilmTable, BVHTree
    from ptina_amd.common import ctx, reset_all
    films = {}
    for lds, tree, wide, quant in ((1, 1, 0, 1), (0, 1, 0, 1), (1, 0, 0, 1), (0, 0, 0, 1), (0, 1, 1, 1), (0, 0, 1, 1), (0, 1, 1, 0), (1, 1, 1, 0), (1, 0, 1, 0)):
        reset_all()
        eng = _engine(None, scenes.scene_s978(), 96, 80, mode='fast')
        c = ctx()
        c.set_option('lds', lds)
        c.set_option('tree', tree)
        c.set_option('wide', wide)
        c.set_option('wide_quant', quant)
        BVHTree().build()
        c.set_option('count', 1)
        c.call('mpt_reset_counters')
        eng.render(6)
        films[(lds, tree, wide, quant)] = (FilmTable().get_raw().copy(), FilmTable().get_image().copy(), c.get_option('last_kernel'),
                                           c.counters())
        if wide:
            assert c.get_option('wide_nodes') > 0 and 1 < c.get_option('wide_depth') <= c.get_option('fast_depth')
    reset_all()
    assert films[(1, 1, 0, 1)][2] == 1 and films[(0, 1, 0, 1)][2] == 0 and films[(0, 1, 1, 1)][2] == 2 and films[(0, 1, 1, 0)][2] == 2
    # the LDS-resident kernel over the 4-wide nodes (exact boxes; what a scene that fits LDS gets by default): the gather kernel's
    # tree and boxes, its own sort (distance bits over 16-bit ids in one word) and the origin triangle filtered in the LEAF step
    assert films[(1, 1, 1, 0)][2] == 5 and films[(1, 0, 1, 0)][2] == 5
    assert_parity(films[(1, 1, 1, 0)][1], films[(0, 1, 1, 0)][1], *FAST, what='4-wide nodes in LDS vs gathered (SAH)')
    assert_parity(films[(1, 1, 1, 0)][1], films[(1, 1, 0, 1)][1], *FAST, what='4-wide nodes in LDS vs binary nodes in LDS (SAH)')
    assert_parity(films[(1, 0, 1, 0)][1], films[(1, 0, 0, 1)][1], *FAST, what='4-wide nodes in LDS vs binary nodes in LDS (LBVH)')
    l4, g4 = films[(1, 1, 1, 0)][3], films[(0, 1, 1, 0)][3]
    assert l4['rays'] == g4['rays'] and np.all(films[(1, 1, 1, 0)][0][:, 3] == 6)
    assert 0.98 * g4['n_node'] <= l4['n_node'] <= 1.02 * g4['n_node']               # same boxes; the order of near-equal entries may differ
    assert 0.98 * g4['n_tri'] <= l4['n_tri'] <= 1.02 * g4['n_tri']                  # (the origin triangle's own LEAF step is not counted: the reference never tests it)
    assert np.array_equal(films[(1, 1, 0, 1)][0], films[(0, 1, 0, 1)][0])
    assert np.array_equal(films[(1, 0, 0, 1)][0], films[(0, 0, 0, 1)][0])
    assert_parity(films[(1, 1, 0, 1)][1], films[(1, 0, 0, 1)][1], *FAST, what='SAH tree vs LBVH')
    assert_parity(films[(0, 1, 1, 1)][1], films[(1, 1, 0, 1)][1], *FAST, what='4-wide nodes vs binary tree (SAH)')
    assert_parity(films[(0, 0, 1, 1)][1], films[(1, 0, 0, 1)][1], *FAST, what='4-wide nodes vs binary tree (LBVH)')
    assert_parity(films[(0, 1, 1, 1)][1], films[(0, 1, 1, 0)][1], *FAST, what='8-bit child boxes vs exact ones (4-wide, SAH)')
    for k in ((0, 1, 1, 1), (0, 0, 1, 1), (0, 1, 1, 0)):
        assert np.all(films[k][0][:, 3] == 6)
    # boxes rounded outwards: the quantised walk visits a few more nodes (measured: +2 %; the order of the children can
    # change with the rounded entry distances, so not strictly a superset)
    exact, quant = films[(0, 1, 1, 0)][3], films[(0, 1, 1, 1)][3]
    assert exact['rays'] == quant['rays']
    assert 0.98 * exact['n_node'] <= quant['n_node'] <= 1.10 * exact['n_node']
    assert 0.98 * exact['n_tri'] <= quant['n_tri'] <= 1.15 * exact['n_tri']


_SPILL_SCRIPT = r'''
import sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + '/tests')
from ptina_amd import scenes
from ptina_amd.common import ctx, reset_all
from ptina_amd.things import FilmTable
from helpers import setup_engine
films = []
for sync_each in (True, False):
    reset_all()
    eng = setup_engine(scenes.scene_s978(), 160, 128, mode='fast')
    c = ctx()
    c.set_option('lds', 0)                 # the 4-wide gather kernel (last_kernel 2)
    c.set_option('batch', 2)
    c.set_option('pipe_depth', 4)          # four launches in flight
    for _ in range(8):
        eng.render(2)                      # one launch each
        if sync_each:
            c.call('mpt_synchronize')
    films.append(FilmTable().get_raw().copy())
    assert c.get_option('last_kernel') == 2 and c.get_option('cur_depth') == 4
assert np.all(films[0][:, 3] == 16)
print('EQUAL' if np.array_equal(films[0], films[1]) else 'DIFFERENT %d' % int((films[0] != films[1]).any(axis=1).sum()))
'''


def test_pipelined_wide_launches_keep_their_own_spill_strips(fresh, tmp_path):
    '''round-2 ADVICE: the overflow strips of the wide kernel's per-lane stacks are indexed by block and lane only, and the
    launches of different ring slots are resident together -- each slot needs its own.  A test build of the library keeps
    only 4 stack levels in LDS (libmiptina_spilltest.so, make -C ptina_amd/csrc spilltest), so every ray of the
    978-triangle scene runs through the strips: eight launches issued back to back (four in flight) must give the film
    of eight launches issued one at a time, bit for bit'''
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, 'ptina_amd', 'libmiptina_spilltest.so')
    assert os.path.exists(lib), 'build it with make -C ptina_amd/csrc spilltest (__graft_entry__.build() does)'
    script = tmp_path / 'spill.py'
    script.write_text(_SPILL_SCRIPT)
    env = dict(os.environ, MIPTINA_LIB=lib)
    r = subprocess.run([sys.executable, str(script), root], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'EQUAL' in r.stdout, r.stdout + r.stderr


def test_pooled_lds_kernel_gives_the_same_film_bit_for_bit(fresh, tmp_path):
    '''option "pool": the LDS-resident kernel with its waves specialised (tracer waves traverse, shader waves run the bounces
    64 at a time) and paths migrating between lanes and waves through two LDS pools at every bounce.  Nothing observable may
    depend on where a path ran: the unspecialised kernel's film up to the last bits (see below) and its work counters, for
    ragged films, several batches, 1 to 5 shader waves, the benchmark scene and a scene with every kind of light and lobe
    (with fewer shader waves more bounces are done by the tracers' own copy of the code, so even the shader count moves last bits)'''
    # The pooled kernel is an A/B build of the library since round 4 (measured 15-50 % slower than the product kernel):
    # libmiptina_pool.so (make -C ptina_amd/csrc pool; __graft_entry__.build() does), loaded by a process of its own
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, 'ptina_amd', 'libmiptina_pool.so')
    assert os.path.exists(lib), 'build it with make -C ptina_amd/csrc pool (__graft_entry__.build() does)'
    from ptina_amd.common import ctx, reset_all
    _engine(None, scenes.scene_s34(), 16, 16, mode='fast')
    with pytest.raises(RuntimeError, match='built without the pooled'):
        ctx().set_option('pool', 1)                # the product library says so instead of silently running another kernel
    reset_all()
    r = subprocess.run([sys.executable, os.path.join(root, 'tests', 'pool_check_script.py'), root], env=dict(os.environ, MIPTINA_LIB=lib),
                       capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'POOL-OK' in r.stdout


def test_shadow_rays_that_cannot_matter_are_not_traced(fresh, oracle_mod):
    '''option "skip_dark" (on in the production build): a shadow ray only decides whether the candidate direct light is added
    (path.py:50-56); when that candidate is exactly zero -- the light behind the surface, a black lobe -- the ray is not traced.
    An exact elimination, shown where arithmetic is exact: in the STRICT build (no contraction) the film is the same BIT FOR
    BIT with the option on and off, with fewer rays and node fetches counted and every other counter equal.  In the production
    build the bounce after a skipped ray starts from another inlined copy of the same code, which -ffp-contract=fast may fuse
    differently: a handful of pixels move in their last bits (measured 39 of 12 288, 2e-6 relative), nothing more; sample counts,
    shades, draws and bounces are equal, and the oracle's film is within the usual bounds.'''
    from helpers import assert_parity, setup_oracle
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    lobes = list(scenes.scene_s34())
    mats = list(lobes[2])
    mats[3] = scenes.material(basecolor=(0.9, 0.95, 1.0), roughness=0.25, transmission=0.8, ior=1.5, specular=0.5)
    mats[4] = scenes.material(basecolor=(0.0, 0.0, 0.0), roughness=0.5, specular=0.0)     # black: many zero candidates
    lobes[2] = mats
    area = np.array([[1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 3.9], [0.0, -1.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    point = np.eye(4)
    point[:3, 3] = (-1.2, 2.5, 1.0)
    lights = [(area, np.array([12.0, 11.0, 9.0]), 0.7, 'AREA'), (point, np.array([20.0, 20.0, 24.0]), 0.3, 'POINT')]
    for scene, lts, nx, ny, spp in ((scenes.scene_s978(), None, 128, 96, 8), (tuple(lobes), lights, 70, 33, 6)):
        for mode, lds in (('strict', 1), ('fast', 1), ('fast', 0)):
            out = {}
            for skip in (1, 0):
                reset_all()
                eng = _engine(None, scene, nx, ny, mode=mode, lights=lts)
                c = ctx()
                c.set_option('lds', lds)
                c.set_option('skip_dark', skip)
                c.set_option('count', 1)
                c.call('mpt_reset_counters')
                eng.render(spp)
                out[skip] = (FilmTable().get_raw().copy(), FilmTable().get_image().copy(), c.counters())
            reset_all()
            a, b = out[1][2], out[0][2]
            assert a['rays'] < b['rays'] and a['n_box'] < b['n_box'] and a['n_tri'] <= b['n_tri']
            assert a['samples'] == b['samples']
            for k in ('n_shade', 'n_draws', 'bounces'):
                # strict: the same paths exactly; production: a direction that differs in its last bit may turn a grazing hit
                # into a miss (measured: 1 shade of 41 299)
                assert a[k] == b[k] if mode == 'strict' else abs(a[k] - b[k]) <= 1e-3 * b[k], k
            diff = (out[1][0].view(np.uint32) != out[0][0].view(np.uint32)).any(axis=1)
            rel = np.abs(out[1][0].astype(np.float64) - out[0][0]) / (np.abs(out[0][0]) + 1e-30)
            print(f'{mode} {nx}x{ny} lds {lds}: rays {a["rays"] / b["rays"]:.3f}, box tests {a["n_box"] / b["n_box"]:.3f} of the run that traces them; '
                  f'{int(diff.sum())} of {len(diff)} pixels differ in some bit, max relative difference {float(np.nanmax(rel)):.1e}')
            # Bit for bit in both builds: the bounce that follows a skipped shadow ray starts in the same block of the shading
            # pass as the one that follows a traced ray (lane_begin_ray), so there is one copy of that arithmetic and nothing
            # for -ffp-contract=fast to fuse two ways.  (While each stage started its own rays the production build showed
            # 39 of 12 288 pixels with last-bit differences here.)
            assert not diff.any()
        if lts is None:     # (the glass scene against the oracle is test_disney_lobes_parity's subject, with its own bounds)
            ref = setup_oracle(oracle_mod, scene, nx, ny, lights=lts)
            ref.render(spp)
            assert_parity(out[1][1], ref.get_image(), *FAST, what=f'skip_dark on vs oracle {nx}x{ny}')


def test_quantised_boxes_far_from_the_origin(fresh, oracle_mod):
    '''the 8-bit child boxes of the 4-wide nodes are offsets from each node's own box, decoded in the kernel as
    q * (scale * inv) + (origin * inv - o * inv): a scene moved 300-500 units away from the origin (coordinates 100 x
    the size of its triangles' boxes) costs that decode precision first.  The rounding margin must hold: both the
    quantised and the exact 4-wide walk against the oracle on the moved scene, and against each other'''
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    from ptina_amd.tools.matrix import translate
    T = np.array([300.0, -200.0, 500.0])
    vertices, mtlids, materials, images = scenes.scene_s978()
    moved = np.array(vertices, dtype=np.float32, copy=True)
    moved[:, 0:3] += T.astype(np.float32)
    scene = (moved, mtlids, materials, images)
    camera = scenes.BENCH_CAMERA @ translate(-T)
    lights = [(translate(np.array([1.0, 2.0, 3.0]) + T), np.array([32.0, 32.0, 32.0]), 0.5, 'POINT')]
    nx, ny, spp = 96, 80, 8
    ref = setup_oracle(oracle_mod, scene, nx, ny, camera=camera, lights=lights)
    ref.render(spp)
    want = ref.get_image()
    got = {}
    for quant in (1, 0):
        reset_all()
        eng = _engine(None, scene, nx, ny, mode='fast', camera=camera, lights=lights)
        c = ctx()
        c.set_option('lds', 0)
        c.set_option('wide_quant', quant)
        eng.render(spp)
        raw = FilmTable().get_raw().reshape(nx, ny, 4)
        assert c.get_option('last_kernel') == 2 and np.all(raw[..., 3] == spp) and np.isfinite(raw).all()
        got[quant] = FilmTable().get_image().copy()
        # f32 positions 500 units out carry 3e-5 of absolute error: pixels along silhouettes flip (measured below 0.3 %)
        assert_parity(got[quant], want, FAST[0], 0.0051, 2.7e-3, what=f'moved scene, 4-wide quant={quant} vs oracle')   # measured 0.39 % / 2.05e-3 (bounds = 1.3 x)
    reset_all()
    assert_parity(got[1], got[0], *FAST, what='moved scene, 8-bit boxes vs exact ones')


def test_launch_pipelining_does_not_change_the_film(fresh):
    '''G launches on 1/G of the CUs each, D batches in flight: same film bit for bit, any G and D,
    including batches of different sizes back to back (the ring of slots is resized in between)'''
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    nx, ny = 96, 80
    films = {}
    for depth, div in ((2, 1), (0, 0), (3, 2), (6, 4), (4, 8)):
        reset_all()
        eng = _engine(None, scenes.scene_s978(), nx, ny, mode='fast')
        c = ctx()
        c.set_option('pipe_depth', depth)
        c.set_option('grid_div', div)
        c.set_option('batch', 8)
        for frames in (8, 8, 3, 8, 1, 8, 8, 8):
            eng.render(frames)
            c.call('mpt_flush')
        films[(depth, div)] = FilmTable().get_raw().copy()
        assert c.get_option('cur_div') == (div if div else 4)      # tiny film: auto picks the smallest launches
    reset_all()
    ref = films[(2, 1)]
    assert np.all(ref.reshape(nx, ny, 4)[..., 3] == 52)
    for key, film in films.items():
        assert np.array_equal(film, ref), key


def test_a_launch_that_finds_the_gpu_idle_takes_the_whole_chip(fresh):
    '''small shares render as G launches on 1/G of the CUs each, so that G of them overlap -- but a step that
    ends with a read-back has one launch in flight at a time: that launch must not be confined to 1/G of the
    chip (a 1/8 share of the benchmark film took 1.69 ms that way instead of 0.90)'''
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx
    eng = _engine(None, scenes.scene_s978(), 512, 512, mode='fast')
    c = ctx()
    c.set_option('batch', 8)
    c.call('mpt_set_stripes', 16, 0, 8)
    for _ in range(3):
        eng.render(8)
        FilmTable().get_image()                     # blocks: nothing is in flight when the next launch is issued
        assert c.get_option('cur_div') == 4         # the ring is laid out for four overlapping launches ...
        assert c.get_option('last_div') == 1        # ... and this one found it idle
    c.set_option('grid_div', 4)                     # an explicit G is honoured as given
    eng.render(8)
    FilmTable().get_image()
    assert c.get_option('last_div') == 4


def test_lights_and_area_light_parity(fresh, oracle_mod):
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.tools.matrix import translate
    rot = np.eye(4)
    rot[:3, :3] = [[1, 0, 0], [0, 0, 1], [0, -1, 0]]        # area light facing down (+z -> -y)
    lights = [(translate([0, 3.9, 0]) @ rot, np.array([18.0, 16.0, 12.0]), 0.6, 'AREA'),
              (translate([-1.2, 3.0, 1.0]), np.array([6.0, 6.0, 9.0]), 0.2, 'POINT')]
    scene = scenes.scene_s34()
    ref = setup_oracle(oracle_mod, scene, 64, 64, lights=lights, world=([0.3, 0.3, 0.4, 1.0], -1))
    ref.render(16)
    for mode, tol in (('strict', 1e-4), ('fast', 1e-3)):
        from ptina_amd.common import reset_all
        reset_all()
        eng = _engine(None, scene, 64, 64, mode=mode, lights=lights, world=([0.3, 0.3, 0.4, 1.0], -1))
        eng.render(16)
        assert_parity(FilmTable().get_image(), ref.get_image(), *bounds(mode), what=f'lights {mode}')



@pytest.mark.parametrize('seed', [11, 12, 13, 14, 15, 16])
def test_random_scenes_parity(fresh, oracle_mod, seed):
    '''seeded random scenes -- the cornell walls plus 5..300 triangles of random place, size and (smooth) normals, random
    opaque Disney materials (every parameter but clearcoat and transmission, whose reference arithmetic is its own subject
    above), 1..3 point / area lights, a random constant world light -- through the strict build and through the production
    build's LDS-resident, binary gather and 4-wide gather kernels, against the oracle at the calibrated bounds.  Nothing here
    is tuned to the benchmark scene: different lane mixes for the in-wave scheduler, paths that end early (dark materials),
    shadow rays that are skipped, lights that are hit directly.'''
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    from ptina_amd.tools.matrix import translate
    rng = np.random.default_rng(seed)
    walls = scenes.cornell_walls()
    k = int(rng.integers(5, 301)) if seed % 2 else int(rng.integers(5, 61))
    c = rng.uniform([-1.6, 0.3, -1.6], [1.6, 3.4, 1.2], (k, 1, 3))
    P = c + rng.normal(0.0, 1.0, (k, 3, 3)) * rng.uniform(0.08, 0.7, (k, 1, 1))
    fn = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    fn /= np.linalg.norm(fn, axis=1, keepdims=True) + 1e-30
    N = fn[:, None, :] + rng.normal(0.0, 0.25, (k, 3, 3))
    N /= np.linalg.norm(N, axis=2, keepdims=True)
    T = rng.uniform(0.0, 1.0, (k, 3, 2))
    nm = int(rng.integers(2, 6))
    M = rng.integers(3, 3 + nm, k).astype(np.int32)
    vertices, mtlids = scenes._compose([walls, (P, N, T, M)])
    mats = list(scenes.WALL_MATERIALS)
    for _ in range(nm):
        mats.append(scenes.material(basecolor=tuple(rng.uniform(0.0, 1.0, 3) * (rng.random() > 0.15)), metallic=float(rng.random() ** 2),
                                    roughness=float(rng.uniform(0.05, 1.0)), specular=float(rng.random()),
                                    specularTint=float(rng.random()), subsurface=float(rng.random() * (rng.random() > 0.5)),
                                    sheen=float(rng.random() * (rng.random() > 0.5)), sheenTint=float(rng.random())))
    scene = (vertices, mtlids, mats, [])
    rot = np.eye(4)
    rot[:3, :3] = [[1, 0, 0], [0, 0, 1], [0, -1, 0]]
    lights = []
    for _ in range(int(rng.integers(1, 4))):
        pos = rng.uniform([-1.5, 2.2, -1.5], [1.5, 3.8, 1.5])
        if rng.random() < 0.5:
            lights.append((translate(list(pos)) @ rot, rng.uniform(4.0, 20.0, 3), float(rng.uniform(0.2, 0.7)), 'AREA'))
        else:
            lights.append((translate(list(pos)), rng.uniform(4.0, 20.0, 3), float(rng.uniform(0.05, 0.4)), 'POINT'))
    world = ([float(x) for x in rng.uniform(0.0, 0.4, 3)] + [1.0], -1)
    nx, ny, spp = 48, 40, 8
    ref = setup_oracle(oracle_mod, scene, nx, ny, lights=lights, world=world)
    ref.render(spp)
    want = ref.get_image()
    assert np.isfinite(want).all()
    for mode, opts in (('strict', {}), ('fast', {}), ('fast', {'lds_wide': 0}), ('fast', {'lds': 0, 'wide': 0}), ('fast', {'lds': 0})):
        reset_all()
        eng = _engine(None, scene, nx, ny, mode=mode, lights=lights, world=world)
        for key, val in opts.items():
            ctx().set_option(key, val)
        eng.render(spp)
        raw = FilmTable().get_raw().reshape(nx, ny, 4)
        assert np.all(raw[..., 3] == spp)
        kernel = ctx().get_option('last_kernel')
        if mode == 'fast':
            assert kernel == (5 if not opts else 1 if 'lds_wide' in opts else 0 if 'wide' in opts else 2), (opts, kernel)
        assert_parity(FilmTable().get_image(), want, *bounds(mode), what=f'random scene {seed} ({k} triangles) {mode} {opts}')
    reset_all()

def test_forty_random_scenes_at_the_stated_bounds(fresh):
    '''the 40 seeded random scenes of tools/stress_random_scenes.py (1 .. 900 triangles beside the walls,
    random opaque materials, lights, film sizes, 1 .. 39 spp, batch sizes) through the strict build and the four production
    kernels, every production film against the strict film at the STATED bounds (99.5 % of the pixels within 1e-3 (1 + |ref|),
    rel-RMSE <= 1e-3 -- no silent factor) with explicit firefly accounting: a pixel may leave the per-pixel bound only by what
    ONE replaced sample explains (the largest single sample of that pixel in either film / spp; the films are rendered a
    frame at a time for that), at most ceil(0.0005 x pixels) pixels per scene may, and the rel-RMSE is taken without them.
    Also: the binary LDS kernel and the binary gather kernel agree bit for bit on every scene'''
    from helpers import stress_compare
    lines, bad, most = [], [], 0
    for seed in range(100, 140):
        ok, flies, msgs = stress_compare(seed, log=lines.append)
        most = max(most, flies)
        if not ok:
            bad.append(seed)
    print('\n'.join(lines))
    print(f'firefly pixels claimed: at most {most} in a scene; scenes failing: {bad}')
    assert not bad, (bad, [l for l in lines if '<<<<' in l or 'more than one' in l])


def test_textures_and_environment_parity(fresh, oracle_mod):
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.common import reset_all
    v, m, mats, _ = scenes.scene_s978()
    rng = np.random.default_rng(5)
    checker = np.ones((8, 8, 3), np.float32)
    checker[::2, 1::2] = 0.2
    checker[1::2, ::2] = 0.2
    mats = [list(x) for x in mats]
    mats[3][0] = ([1.0, 0.9, 0.8], 1)                      # basecolor textured by image 1
    mats[3][2] = (0.9, 2)                                   # roughness textured by image 2 (grey)
    images = [scenes.env_image(64, 32), checker, rng.uniform(0.3, 1.0, (5, 7)).astype(np.float32)]
    scene = (v, m, mats, images)
    world = ([1.0, 1.0, 1.0, 1.0], 0)
    ref = setup_oracle(oracle_mod, scene, 64, 64, world=world)
    ref.render(16)
    for mode, tol in (('strict', 1e-4), ('fast', 1e-3)):
        reset_all()
        eng = _engine(None, scene, 64, 64, mode=mode, world=world)
        eng.render(16)
        assert_parity(FilmTable().get_image(), ref.get_image(), *bounds(mode), what=f'textures {mode}')


def test_every_material_parameter_textured(fresh, oracle_mod):
    '''ParameterPair.get (mtllib.py:30-38) for each of the twelve parameters: factor x bilinear texel of
    its own image, scalar parameters taking .x; texture coordinates run from -1.3 to 2.4 so the
    wrap-around addressing of image.py:137-148 (Python floor-mod on negative texel indices) is used'''
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.common import reset_all
    v, m, mats, _ = scenes.scene_s34()
    v = v.copy()
    v[:, 6:8] = v[:, 6:8] * 3.7 - 1.3
    rng = np.random.default_rng(77)
    images = [rng.uniform(0.2, 1.0, (4 + k, 9 - (k % 5), 3 if k % 3 == 0 else 1)).astype(np.float32) for k in range(12)]
    fac = scenes.material(basecolor=(0.9, 0.8, 0.7), metallic=0.6, roughness=0.8, specular=0.9, specularTint=0.9,
                          subsurface=0.8, sheen=0.9, sheenTint=0.9, clearcoat=0.0, clearcoatGloss=0.9,
                          transmission=0.0, ior=1.6)
    textured = [(f, k) for k, (f, _) in enumerate(fac)]
    mats = list(mats)
    mats[0] = textured            # white walls
    mats[3] = textured            # tall box
    scene = (v, m, mats, images)
    ref = setup_oracle(oracle_mod, scene, 64, 64)
    ref.render(16)
    for mode, tol in (('strict', 1e-4), ('fast', 1e-3)):
        reset_all()
        eng = _engine(None, scene, 64, 64, mode=mode)
        eng.render(16)
        assert_parity(FilmTable().get_image(), ref.get_image(), *bounds(mode), what=f'12 textured parameters {mode}')
    reset_all()


def test_no_lights_default_material_world_only(fresh, oracle_mod):
    '''LightPool().clear() (count 0: the light triple is still drawn, path.py:48, light/__init__.py:117-121),
    ModelPool.load(vertices) without material ids (-1: the default material of mtllib.py:82-93) and a
    bright constant world light as the only source'''
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.common import reset_all
    v, m, mats, _ = scenes.scene_s978()
    scene = (v, None, mats, [])
    world = ([0.9, 1.0, 1.1, 1.0], -1)
    ref = setup_oracle(oracle_mod, scene, 64, 64, lights=[], world=world)
    ref.render(16)
    want = ref.get_image()
    assert want[..., :3].mean() > 0.05
    for mode, tol in (('strict', 1e-4), ('fast', 1e-3)):
        reset_all()
        eng = _engine(None, scene, 64, 64, mode=mode, lights=[], world=world)
        eng.render(16)
        assert_parity(FilmTable().get_image(), want, *bounds(mode), what=f'world only {mode}')
    reset_all()


def test_preview_aov_parity(fresh, oracle_mod):
    '''PreviewEngine (engine/preview.py:18-41): albedo (textured base colour) and shading normal of the
    primary hit into passes 1 and 2, both builds'''
    from helpers import setup_oracle
    from ptina_amd.things import FilmTable
    from ptina_amd.common import reset_all
    from ptina_amd.engine.preview import PreviewEngine
    v, m, mats, _ = scenes.scene_s978()
    checker = np.ones((8, 8, 3), np.float32)
    checker[::2, 1::2] = 0.2
    checker[1::2, ::2] = 0.2
    mats = [list(x) for x in mats]
    mats[3][0] = ([1.0, 0.9, 0.8], 0)
    scene = (v, m, mats, [checker])
    ref = setup_oracle(oracle_mod, scene, 48, 48)
    ref.render_preview()
    ref.render_preview()
    for mode in ('strict', 'fast'):
        reset_all()
        _engine(None, scene, 48, 48, mode=mode)
        PreviewEngine().render()
        PreviewEngine().render()
        for p in (1, 2):
            a, b = FilmTable().get_image(p), ref.get_image(p)
            close = np.isclose(a, b, rtol=1e-4, atol=1e-5).all(axis=-1)
            assert close.mean() > 0.99, (mode, p, close.mean())
        assert np.all(FilmTable().get_raw(0) == 0)             # path pass untouched
    reset_all()


def test_edge_cases_empty_and_single_triangle(fresh, oracle_mod):
    from helpers import setup_oracle
    from ptina_amd.things import FilmTable
    from ptina_amd.common import reset_all
    tri = np.zeros((3, 8), np.float32)
    tri[:, :3] = [[-1, 0, 0], [1, 0, 0], [0, 2, 0]]
    tri[:, 5] = 1
    for verts, ids in ((np.zeros((0, 8), np.float32), np.zeros(0, np.int32)), (tri, np.array([-1], np.int32))):
        scene = (verts, ids, [], [])
        ref = setup_oracle(oracle_mod, scene, 32, 32)
        ref.render(2)
        for mode in ('strict', 'fast'):
            reset_all()
            eng = _engine(None, scene, 32, 32, mode=mode)
            eng.render(2)
            a, b = FilmTable().get_image(), ref.get_image()
            # every ray misses (one face: the reference never writes the root box)
            assert np.allclose(a, b, rtol=1e-5, atol=1e-6)
    reset_all()


def test_tiny_scenes_through_every_production_kernel(fresh, oracle_mod):
    '''two and three triangles (one and two internal nodes: a 4-wide record with two or three of its four slots
    used) and the 34-triangle box, through the LDS-resident kernel, the binary gather kernel and the 4-wide one
    with 8-bit and with exact child boxes: all against the oracle'''
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    quad = np.zeros((9, 8), np.float32)
    quad[:, :3] = [[-1.5, 0.2, 0], [1.5, 0.2, 0], [1.5, 2.8, 0], [-1.5, 0.2, 0], [1.5, 2.8, 0], [-1.5, 2.8, 0],
                   [-0.5, 1.0, 0.8], [0.7, 1.2, 0.9], [0.1, 2.2, 0.6]]
    quad[:, 5] = 1
    scenes_ = {'two triangles': (quad[:6].copy(), np.array([-1, -1], np.int32), [], []),
               'three triangles': (quad.copy(), np.array([-1, -1, -1], np.int32), [], []),
               's34': scenes.scene_s34()}
    for name, scene in scenes_.items():
        ref = setup_oracle(oracle_mod, scene, 48, 40)
        ref.render(8)
        want = ref.get_image()
        for lds, wide, quant, kernel in ((1, 1, 1, 5), (1, 0, 1, 1), (0, 0, 1, 0), (0, 1, 1, 2), (0, 1, 0, 2)):
            reset_all()
            eng = _engine(None, scene, 48, 40, mode='fast')
            c = ctx()
            c.set_option('lds', lds)
            c.set_option('wide', wide)
            c.set_option('wide_quant', quant)
            eng.render(8)
            raw = FilmTable().get_raw().reshape(48, 40, 4)
            assert c.get_option('last_kernel') == kernel and np.all(raw[..., 3] == 8), (name, lds, wide, quant)
            assert_parity(FilmTable().get_image(), want, *FAST, what=f'{name}: lds={lds} wide={wide} quant={quant}')
    reset_all()


@pytest.mark.parametrize('k', [0.002, 50.0])
def test_scene_scale_dependence_is_the_references(fresh, oracle_mod, k):
    '''eps = 1e-6 and inf = 1e6 are absolute (common.py:32-33) and Face.intersect tests |n.d| against eps
    with the UNNORMALISED normal (geometries.py:123-129): the same scene scaled by k is not
    the same image.  Both builds must follow the oracle at either end; a degenerate (zero-area) and a
    needle triangle ride along.'''
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.common import reset_all
    from ptina_amd.tools.matrix import translate, scale
    v, m, mats, _ = scenes.scene_s34()
    extra = np.zeros((6, 8), np.float32)
    extra[0:3, :3] = [[0.5, 1.0, 0.5], [0.5, 1.0, 0.5], [0.9, 1.4, 0.5]]            # two coincident vertices
    extra[3:6, :3] = [[-1.5, 0.5, 1.0], [-1.5, 3.5, 1.0], [-1.5 + 1e-4, 2.0, 1.0]]   # needle
    extra[:, 3:6] = [0, 0, 1]
    v = np.concatenate([v, extra]).astype(np.float32)
    m = np.concatenate([m, [1, 2]]).astype(np.int32)
    v[:, :3] *= k
    cam = scenes.BENCH_CAMERA @ scale(1.0 / k)
    lights = [(translate([1.0 * k, 2.0 * k, 3.0 * k]), np.array([32.0, 32.0, 32.0]), 0.5 * k, 'POINT')]
    scene = (v, m, mats, [])
    ref = setup_oracle(oracle_mod, scene, 64, 64, camera=cam, lights=lights)
    ref.render(8)
    want = ref.get_image()
    assert np.isfinite(want).all() and want[..., :3].mean() > 0.1
    for mode, tol in (('strict', 1e-4), ('fast', 1e-3)):
        reset_all()
        eng = _engine(None, scene, 64, 64, mode=mode, camera=cam, lights=lights)
        eng.render(8)
        if k < 1:
            # a scene 500x smaller than the constants eps = 1e-6, inf = 1e6 were chosen for: many hits sit on the
            # |b| >= eps threshold of geometries.py:129 and flip with the last bit (measured: strict 0.12 % of the
            # pixels / rel-RMSE 1.88e-3, fast 0.59 % / 3.15e-3; the bounds are 1.3 x that)
            b = (1e-4, 0.0016, 2.45e-3) if mode == 'strict' else (1e-3, 0.0077, 4.1e-3)
        else:
            b = bounds(mode)
        assert_parity(FilmTable().get_image(), want, *b, what=f'scale {k} {mode}')
    reset_all()


def test_worker_facade_through_the_thread_proxy(fresh):
    '''the Blender add-on's view of the engine: every function of worker.py:11-87 called through
    tools.mtworker's daemon thread (blender.py:565-580), against the same scene driven directly'''
    import threading
    from ptina_amd.tools.mtworker import DaemonModule, OnDemandProxy
    from ptina_amd.tools.matrix import translate
    from ptina_amd.common import reset_all
    from ptina_amd.things import FilmTable
    v, m, mats, _ = scenes.scene_s34()
    checker = np.ones((4, 4, 3), np.float32)
    checker[::2, 1::2] = 0.3
    mats = [list(x) for x in mats]
    mats[3][0] = ([1.0, 1.0, 1.0], 0)
    light = (translate([0.5, 3.2, 0.5]), np.array([20.0, 18.0, 16.0]), 0.3, 'POINT')
    world = ([0.2, 0.2, 0.3, 1.0], -1)
    nx, ny, spp = 40, 28, 3

    eng = _engine(None, (v, m, mats, [checker]), nx, ny, mode='fast', lights=[light], world=world)
    eng.render(spp)
    from ptina_amd.engine.preview import PreviewEngine
    PreviewEngine().render()
    direct = [FilmTable().get_raw(p).copy() for p in range(3)]
    reset_all()

    seen = []

    def module():
        seen.append(threading.get_ident())
        from ptina_amd import worker
        return worker
    w = OnDemandProxy(lambda: DaemonModule(module))
    w.init()
    w.set_size(nx, ny)
    assert w.get_size() == (nx, ny)
    w.load_model(v, m)
    w.load_materials(mats)
    w.load_images([checker])
    w.build_tree()
    w.set_camera(scenes.BENCH_CAMERA)
    w.clear_lights()
    w.add_light(*light)
    w.set_world_light(*world)
    w.set_mlt_param(0.3, 0.03)                       # accepted, no effect on the path engine (worker.py:45-49)
    for _ in range(spp):
        w.render()
    w.render_preview()
    w.synchronize()
    img = w.get_image()
    assert img.shape == (nx, ny, 4) and seen and seen[0] != threading.get_ident()
    flat = np.zeros(nx * ny * 3, np.float32)
    w.fast_export_image(flat, 0)
    assert np.allclose(flat.reshape(ny, nx, 3), np.swapaxes(img[..., :3], 0, 1), rtol=1e-6, atol=1e-7)
    got = [w.get_image(p) for p in range(3)]
    for p in range(3):
        want = direct[p].reshape(nx, ny, 4)
        res = np.where(want[..., 3:] != 0, want / np.where(want[..., 3:] != 0, want[..., 3:], 1), [0.9, 0.4, 0.9, 0.0])
        assert np.allclose(got[p][..., :3], res[..., :3], rtol=1e-6, atol=1e-7), p
    w.clear()
    assert np.all(w.get_image()[..., 3] == 0)        # FilmTable.clear: every pass (filmtable.py:44-45)
    assert np.all(w.get_image(1)[..., 3] == 0)
    reset_all()


def test_errors_are_loud(fresh):
    from ptina_amd.things import init_things, FilmTable, ModelPool, BVHTree
    from ptina_amd.engine.path import PathEngine
    init_things(max_filmsize=64 * 64)
    eng = PathEngine()
    with pytest.raises(RuntimeError, match='film size'):
        eng.render()
    with pytest.raises(RuntimeError, match='max_filmsize'):
        FilmTable().set_size(128, 128)
    FilmTable().set_size(32, 32)
    with pytest.raises(RuntimeError, match='BVH not built'):
        eng.render()
    ModelPool().load(scenes.scene_s34()[0], scenes.scene_s34()[1])
    with pytest.raises(RuntimeError, match='BVH not built'):
        eng.render()
    BVHTree().build()
    eng.render()
    assert np.all(FilmTable().get_raw()[:, 3] == 1)
    from ptina_amd.common import ctx
    with pytest.raises(RuntimeError, match='multiple of 16'):
        ctx().call('mpt_set_stripes', 8, 0, 2)
    with pytest.raises(RuntimeError, match='stripe index'):
        ctx().call('mpt_set_stripes', 16, 2, 2)
    with pytest.raises(RuntimeError, match='slab'):
        ctx().call('mpt_set_slab', 10, 40)
    with pytest.raises(RuntimeError, match='unknown option'):
        ctx().set_option('no_such_option', 1)
    with pytest.raises(RuntimeError, match='grid_div'):
        ctx().set_option('grid_div', 9)
    ctx().call('mpt_set_stripes', 16, 1, 2)       # a share of [16, 32)
    FilmTable().clear()
    eng.render()
    w = FilmTable().get_raw().reshape(32, 32, 4)[..., 3]
    assert np.all(w[16:] == 1) and np.all(w[:16] == 0)


def test_config3_film_size_and_a_stripe_share(fresh, oracle_mod):
    '''BASELINE configs[2]'s film (2048x2048, needs max_filmsize = 2^22): every pixel gets its
    samples, the share of one rank of eight is exactly its stripes, and a window agrees with the oracle'''
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    from ptina_amd.dist import stripe_columns
    n, spp = 2048, 2
    scene = scenes.scene_s978()
    eng = _engine(None, scene, n, n, mode='fast', max_filmsize=n * n)
    eng.render(spp)
    raw = FilmTable().get_raw().reshape(n, n, 4)
    assert np.all(raw[..., 3] == spp) and np.isfinite(raw).all() and raw[..., :3].min() >= 0
    img = FilmTable().get_image()
    x0, x1 = 1000, 1004
    ref = setup_oracle(oracle_mod, scene, n, n)
    ref.set_window(x0, x1)
    ref.render(spp)
    assert_parity(img[x0:x1], ref.get_image()[x0:x1], *FAST, what='2048x2048 window')
    reset_all()
    eng = _engine(None, scene, n, n, mode='fast', max_filmsize=n * n)
    ctx().call('mpt_set_stripes', 16, 5, 8)
    eng.render(spp)
    part = FilmTable().get_raw().reshape(n, n, 4)
    cols = stripe_columns(n, 8, 5)
    assert len(cols) == n // 8
    assert np.array_equal(part[cols], raw[cols])
    assert np.all(part[np.setdiff1d(np.arange(n), cols)] == 0)
    reset_all()


@pytest.mark.parametrize('name', ['s978', 's34'])
def test_full_size_properties_and_oracle_parity(fresh, oracle_mod, name):
    '''BASELINE configs[1] and configs[0] at full size (512x512x32, S978 / S34): size-independent properties, run-to-run
    bit reproducibility, fast-vs-strict agreement, and oracle parity on the WHOLE film (the oracle does
    the 8.4 M samples in a few seconds on the GPU box's host cores; on fewer than 8 cores it falls back
    to a window of columns)'''
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.common import reset_all
    scene = scenes.get_scene(name)
    nx = ny = 512
    spp = 32
    imgs = {}
    for key, mode in (('fast', 'fast'), ('fast2', 'fast'), ('strict', 'strict')):
        reset_all()
        eng = _engine(None, scene, nx, ny, mode=mode)
        img = _bench_sequence(eng, FilmTable(), spp)
        raw = FilmTable().get_raw()
        assert np.all(raw[:, 3] == spp) and np.isfinite(raw).all() and raw[:, :3].min() >= 0
        imgs[key] = img
    reset_all()
    assert np.array_equal(imgs['fast'], imgs['fast2']), 'render is not run-to-run deterministic'
    assert_parity(imgs['fast'], imgs['strict'], *FAST, what='full-size fast vs strict')
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    x0, x1 = (0, nx) if cores >= 8 else (250, 258)
    ref = setup_oracle(oracle_mod, scene, nx, ny, threads=min(cores, 16))
    ref.set_window(x0, x1)
    ref.render(1)
    ref.clear()
    ref.render(spp)
    want = ref.get_image()[x0:x1]
    # rel-RMSE of the whole 262 144-pixel film: a handful of flipped 32-spp pixels put it at 9e-5 (measured)
    assert_parity(imgs['strict'][x0:x1], want, STRICT[0], STRICT[1], 1.2e-4, what=f'full-size strict, columns [{x0},{x1})')   # measured 9.07e-5 on s978 (9.1e-7 on s34): 1.3 x
    assert_parity(imgs['fast'][x0:x1], want, *FAST, what=f'full-size fast, columns [{x0},{x1})')


def test_rccl_film_gather_single_rank(fresh):
    '''the RCCL path end to end with one rank: dlopen librccl, unique id, communicator, slab,
    gather (a no-op for one rank), barrier and max all-reduce on the render stream'''
    from ptina_amd.things import FilmTable
    from ptina_amd.dist import RcclFilm, slab_bounds
    eng = _engine(None, scenes.scene_s34(), 48, 32)
    comm = RcclFilm(rank=0, world=1)
    assert comm.set_slab(48) == slab_bounds(48, 1, 0) == (0, 48)
    eng.render(2)
    comm.gather(0, 0)
    comm.barrier()
    assert comm.allreduce_max(3.25) == 3.25
    assert np.all(FilmTable().get_raw()[:, 3] == 2)
    comm.close()


@pytest.mark.parametrize('n', [2, 3, 1000, 60000])
def test_gpu_lbvh_build_equals_host_build(fresh, n):
    '''the on-device build (lbvh_build.hip: Morton keys, radix sort, Karras hierarchy, atomic
    bottom-up boxes) is node-for-node the host build, duplicate Morton codes included'''
    from ptina_amd.things import init_things, ModelPool, BVHTree
    from ptina_amd.common import ctx
    v, m, _, _ = scenes.scene_random_tris(n, seed=n, edge=0.05)
    if n >= 1000:
        v[3 * 7:3 * 9] = v[3 * 5:3 * 7]            # exact duplicates -> equal Morton codes
    init_things()
    ModelPool().load(v, m)
    trees = {}
    for gpu in (1, 0):
        ctx().set_option('gpu_build', gpu)
        BVHTree().build()
        trees[gpu] = BVHTree().to_numpy()
    for k in ('mc', 'leaf', 'child', 'bmin', 'bmax', 'depth'):
        assert np.array_equal(trees[0][k], trees[1][k]), k
    assert sorted(trees[1]['leaf']) == list(range(n))


def _wide_records(c, n):
    import ctypes as C
    from ptina_amd import _lib
    nw = C.c_int(0)
    c.call('mpt_get_wide', None, None, 0, C.byref(nw))
    w = np.zeros((max(nw.value, 1), 8, 4), np.float32)
    q = np.zeros((max(nw.value, 1), 4, 4), np.float32)
    c.call('mpt_get_wide', _lib.fptr(w), _lib.fptr(q), nw.value, C.byref(nw))
    return w[:nw.value].view(np.uint32), q[:nw.value].view(np.uint32)


@pytest.mark.parametrize('name,kw', [('s978', {}), ('c5', {'n': 60000}), ('s34', {})])
def test_device_wide_collapse_equals_the_host_pass(fresh, name, kw):
    '''the 4-wide collapse the gather kernels walk (128-byte records with exact boxes, 64-byte records with 8-bit boxes
    rounded outwards), built on the device level by level (wide_build.hip) == the round-2 host pass over downloaded
    records, byte for byte, on the SAH tree and on the plain LBVH; unused child slots name the NaN triangle of slot n'''
    from ptina_amd.things import BVHTree
    from ptina_amd.common import ctx, reset_all
    scene = scenes.get_scene(name, **kw)
    n = scene[1].shape[0]
    for tree in (1, 0):
        recs = {}
        for dev in (1, 0):
            reset_all()
            _engine(None, scene, 16, 16, mode='fast', max_faces=max(n + 1, 1 << 21))
            c = ctx()
            c.set_option('tree', tree)
            c.set_option('wide_build', dev)
            BVHTree().build()
            recs[dev] = _wide_records(c, n) + (c.get_option('wide_nodes'), c.get_option('wide_depth'), c.get_option('wide_ratio_permille'))
        reset_all()
        assert recs[1][2] == recs[0][2] > 0 and recs[1][3] == recs[0][3]
        assert abs(recs[1][4] - recs[0][4]) <= 1
        assert np.array_equal(recs[1][0], recs[0][0]), f'{name} tree {tree}: exact-box records differ'
        assert np.array_equal(recs[1][1], recs[0][1]), f'{name} tree {tree}: quantised records differ'
        ids = recs[1][0][:, 6, :].view(np.int32)
        nw = recs[1][2]
        assert ((ids < nw) & (ids >= -(n + 1))).all() and (ids != 0).all()         # nobody's child is the root
        assert np.array_equal(np.sort(ids[ids > 0]), np.arange(1, nw))             # every wide node but the root has one parent
        leaves = ~ids[(ids < 0) & (ids != ~n)]
        assert np.array_equal(np.sort(leaves), np.arange(n))                       # every triangle in exactly one slot


def test_octant_ordered_8wide_tree_and_kernel(fresh, oracle_mod, tmp_path):
    '''option "wide8" (VERDICT r03 next #3): the fast tree collapsed 8-wide with octant-ordered child slots and 8-bit boxes
    (oct_build.cpp), walked without a sort by render_kernel_oct.  Structure: every 8-wide node but the root is the child of
    exactly one slot, internal children and leaf triangles are numbered consecutively in slot order, every triangle sits in
    exactly one leaf slot (the permutation is one), every child's quantised box holds the boxes of everything below it (checked
    from the leaves up), empty slots are inverted boxes.  Films: 60 000 random triangles and the benchmark scene forced off LDS,
    against the 4-wide kernel (the same hits; equal-depth ties may be met in another order) and against the oracle.
    Since round 5 the 8-wide tree and kernel are an A/B build of the library (measured 17-19 % slower than the 4-wide node):
    libmiptina_oct.so (make -C ptina_amd/csrc oct; __graft_entry__.build() does), loaded by a process of its own
    (tests/oct_check_script.py); the product library refuses the option loudly'''
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, 'ptina_amd', 'libmiptina_oct.so')
    assert os.path.exists(lib), 'build it with make -C ptina_amd/csrc oct (__graft_entry__.build() does)'
    from ptina_amd.common import ctx, reset_all
    _engine(None, scenes.scene_s34(), 16, 16, mode='fast')
    with pytest.raises(RuntimeError, match='built without the 8-wide'):
        ctx().set_option('wide8', 1)
    reset_all()
    r = subprocess.run([sys.executable, os.path.join(root, 'tests', 'oct_check_script.py'), root], env=dict(os.environ, MIPTINA_LIB=lib),
                       capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'OCT-OK' in r.stdout


def _sah_tree(c, n, dev):
    from ptina_amd.things import BVHTree
    c.set_option('sah_build', dev)
    BVHTree().build()
    assert c.get_option('sah_fallback') == 0
    w, q = _wide_records(c, n)
    return w, q, c.get_option('fast_depth'), c.get_option('wide_nodes')


def _check_wide_tree(w, n, nw):
    ids = w[:, 6, :].view(np.int32)
    assert nw > 0
    assert np.array_equal(np.sort(ids[ids > 0]), np.arange(1, nw)), 'every wide node but the root has exactly one parent'
    assert np.array_equal(np.sort(~ids[(ids < 0) & (ids != ~n)]), np.arange(n)), 'every triangle sits in exactly one slot'
    # every child box lies inside ... the boxes of a leaf child are its triangle's (checked against the model by the films);
    # here: finite, lo <= hi for every used slot
    lo = w[:, 0:6:2, :].view(np.float32)
    hi = w[:, 1:6:2, :].view(np.float32)
    used = np.broadcast_to((ids != ~n)[:, None, :], lo.shape)
    assert np.all(np.isfinite(lo[used])) and np.all(np.isfinite(hi[used])) and np.all(lo[used] <= hi[used])


@pytest.mark.parametrize('n', [2, 3, 4, 7, 33, 64, 65, 129, 300, 511, 512, 513, 700, 1023, 1024])
def test_device_sah_finish_kernel_is_the_host_pass_node_for_node(fresh, n):
    '''sah_build.hip, finish kernel: a range of at most 1024 triangles is built by one wave in LDS with the host pass's exact sweep
    (every split of every axis, sorted by (centre, slot); lowest cost, then lowest axis, then lowest split; an axis along which
    the centres do not differ is skipped; 8 positions per lane up to 512 triangles, 16 above).  For n <= 1024 the whole tree is
    that kernel's: the same records as the host pass's,
    byte for byte (through the 4-wide collapse, which is a function of the binary records), the same depth.  With exact
    duplicates in the model (equal centres: ties go to the lower slot)'''
    from ptina_amd.things import init_things, ModelPool
    from ptina_amd.common import ctx
    v, m, _, _ = scenes.scene_random_tris(n, seed=1000 + n, edge=0.2)
    if n >= 7:
        v[3 * 4:3 * 6] = v[3 * 2:3 * 4]            # exact duplicates
    init_things()
    ModelPool().load(v, m)
    c = ctx()
    dev = _sah_tree(c, n, 1)
    host = _sah_tree(c, n, 0)
    assert dev[2] == host[2] and dev[3] == host[3]
    assert np.array_equal(dev[0], host[0]) and np.array_equal(dev[1], host[1])
    _check_wide_tree(dev[0], n, dev[3])


@pytest.mark.parametrize('n', [1025, 1100, 2049, 2100, 5000, 20000, 99382])
def test_device_sah_pass_is_a_valid_deterministic_tree(fresh, n):
    '''sah_build.hip above 1024 triangles: binned levels (chunks, plan, stable scatter) down to ranges of <= 1024, then the
    finish kernels.  A valid tree over the same leaf slots at sizes around the switch, one / several chunks per segment and the size of
    BASELINE config 4; built twice: the same bytes; depth within the stack; surface-area cost (sum of the areas of the
    boxes that are fetched) no worse than 1.1 x the host pass's'''
    from ptina_amd.things import init_things, ModelPool
    from ptina_amd.common import ctx
    v, m, _, _ = scenes.scene_random_tris(n, seed=n, edge=0.05)
    v[3 * 7:3 * 9] = v[3 * 5:3 * 7]
    init_things(max_faces=n + 1)
    ModelPool().load(v, m)
    c = ctx()
    a = _sah_tree(c, n, 1)
    b = _sah_tree(c, n, 1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    _check_wide_tree(a[0], n, a[3])
    assert 2 < a[2] <= 62

    def cost(w):
        lo = w[:, 0:6:2, :].view(np.float32).astype(np.float64)
        hi = w[:, 1:6:2, :].view(np.float32).astype(np.float64)
        ids = w[:, 6, :].view(np.int32)
        d = np.maximum(hi - lo, 0.0)
        area = d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]
        return float(area[ids != ~n].sum())
    h = _sah_tree(c, n, 0)
    print(f'n {n}: depth {a[2]} / host {h[2]}, wide nodes {a[3]} / {h[3]}, area cost {cost(a[0]) / cost(h[0]):.4f} x the host pass')
    assert cost(a[0]) <= 1.1 * cost(h[0])


@pytest.mark.parametrize('kind', ['identical', 'on_a_line', 'huge', 'two_clusters'])
def test_device_sah_pass_on_degenerate_models(fresh, kind):
    '''sah_build.hip where no split can be chosen by cost: 3000 copies of ONE triangle (every centre equal: the binned levels halve
    the ranges by position, the finish kernel in the order of axis 0 -- ties by slot), triangles whose centres differ along x only
    (two axes never take part), coordinates around 1e18 (areas overflow to infinity: no finite cost, the halving fallback), and two
    far-apart clusters of identical triangles (one bin boundary separates everything, then nothing does).  A valid tree every
    time, depth within the stack, built twice: the same bytes; the film finite and every pixel counted'''
    from ptina_amd.things import init_things, ModelPool, FilmTable, MaterialPool, ImagePool, Camera
    from ptina_amd.engine.path import PathEngine
    from ptina_amd.common import ctx
    n = 3000
    v, m, mats, _ = scenes.scene_random_tris(n, seed=7, edge=0.05)
    v = v.reshape(n, 3, 8).copy()
    if kind == 'identical':
        v[:] = v[0]
    elif kind == 'on_a_line':
        v[:, :, 1:3] = v[0, :, 1:3]
        v[:, :, 0] = v[0, :, 0] + np.arange(n, dtype=np.float32)[:, None] * 1e-3
    elif kind == 'huge':
        v[:, :, :3] *= np.float32(1e18)
    else:
        v[:] = v[0]
        v[n // 3:, :, :3] += np.float32(5.0)
    v = v.reshape(n * 3, 8)
    init_things(max_faces=n + 1)
    eng = PathEngine()
    FilmTable().set_size(24, 16)
    ModelPool().load(v, m)
    MaterialPool().load(mats)
    ImagePool().load([])
    c = ctx()
    a = _sah_tree(c, n, 1)
    b = _sah_tree(c, n, 1)
    assert np.array_equal(a[0], b[0]) and a[2] == b[2]
    ids = a[0][:, 6, :].view(np.int32)
    assert np.array_equal(np.sort(ids[ids > 0]), np.arange(1, a[3]))
    assert np.array_equal(np.sort(~ids[(ids < 0) & (ids != ~n)]), np.arange(n))
    assert 2 < a[2] <= 62
    Camera().set_perspective(scenes.BENCH_CAMERA)
    eng.render(2)
    raw = FilmTable().get_raw()
    assert np.all(raw[:, 3] == 2) and np.isfinite(raw).all()
    print(f'{kind}: depth {a[2]}, {a[3]} wide nodes')


def test_device_sah_pass_builds_a_tree_as_good_as_the_host_pass(fresh, oracle_mod):
    '''the SAH re-partition on the device (sah_build.hip: binned above 32 triangles, the host pass's exact sweep below)
    against the round-2 host pass (exact sweep up to 8192) on 60 000 random triangles: a valid tree over the same leaf
    slots (every triangle in exactly one slot of the collapse, DFS numbering, depth within the stack), node fetches and
    triangle tests per ray within 5 % of the host tree's, the film within the FAST bounds of the host tree's film and of the
    oracle's on a window; run twice: the same bytes (deterministic)'''
    from helpers import assert_parity, setup_oracle
    from ptina_amd.things import FilmTable, BVHTree
    from ptina_amd.common import ctx, reset_all
    scene = scenes.get_scene('c5', n=60000)
    n = scene[1].shape[0]
    nx, ny, spp = 192, 160, 4
    out = {}
    for dev in (1, 0, 1):
        reset_all()
        eng = _engine(None, scene, nx, ny, mode='fast', max_faces=n + 1)
        c = ctx()
        c.set_option('sah_build', dev)
        BVHTree().build()
        w, q = _wide_records(c, n)
        ids = w[:, 6, :].view(np.int32)
        nw = c.get_option('wide_nodes')
        assert nw > 0 and 2 < c.get_option('fast_depth') <= 62
        assert np.array_equal(np.sort(ids[ids > 0]), np.arange(1, nw))
        assert np.array_equal(np.sort(~ids[(ids < 0) & (ids != ~n)]), np.arange(n))
        c.set_option('count', 1)
        c.call('mpt_reset_counters')
        eng.render(spp)
        cnt = c.counters()
        key = ('dev2' if dev and 'dev' in out else 'dev') if dev else 'host'
        out[key] = (FilmTable().get_image().copy(), cnt['n_node'] / cnt['rays'], cnt['n_tri'] / cnt['rays'], w.copy(), c.get_option('last_kernel'))
    reset_all()
    assert out['dev'][4] == out['host'][4] == 2
    assert np.array_equal(out['dev'][3], out['dev2'][3]) and np.array_equal(out['dev'][0], out['dev2'][0])
    print(f'node fetches per ray: device SAH {out["dev"][1]:.2f}, host SAH {out["host"][1]:.2f}; triangle tests {out["dev"][2]:.2f} / {out["host"][2]:.2f}')
    assert out['dev'][1] <= 1.05 * out['host'][1] and out['dev'][2] <= 1.05 * out['host'][2]
    assert_parity(out['dev'][0], out['host'][0], *FAST, what='device SAH tree vs host SAH tree')
    ref = setup_oracle(oracle_mod, scene, nx, ny)
    ref.set_window(80, 96)
    ref.render(spp)
    assert_parity(out['dev'][0][80:96], ref.get_image()[80:96], *FAST, what='device SAH tree vs oracle (16 columns)')


def test_render_does_not_depend_on_where_the_tree_was_built(fresh):
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    films = []
    for gpu in (1, 0):
        reset_all()
        from ptina_amd.things import init_things
        init_things()
        ctx().set_option('gpu_build', gpu)
        eng = _engine(None, scenes.scene_s978(), 48, 40, mode='fast')
        eng.render(4)
        films.append(FilmTable().get_raw())
    reset_all()
    assert np.array_equal(films[0], films[1])


def test_ptina_named_driver_script_runs_unchanged(fresh, tmp_path, monkeypatch):
    '''a driver written against PTina's own module names (`from ptina.things import *`, readgltf of
    assets/monkey_cornell.gltf, the pools, BVHTree, Camera, PathEngine, FilmTable) runs as is'''
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('make_assets', os.path.join(root, 'tools', 'make_assets.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(str(tmp_path / 'assets'))
    monkeypatch.chdir(tmp_path)
    script = """
from ptina.things import *
from ptina.engine.path import *
from ptina.tools.readgltf import readgltf
import time

ti.init(ti.cuda)
init_things()
PathEngine()
FilmTable().set_size(128, 128)
vertices, mtlids, materials, images = readgltf('assets/monkey_cornell.gltf')
ModelPool().load(vertices, mtlids)
MaterialPool().load(materials)
ImagePool().load(images)
BVHTree().build()
Camera().set_perspective(np.array(CAMERA))
PathEngine().render()
FilmTable().get_image()
FilmTable().clear()
for i in range(8):
    PathEngine().render()
img = FilmTable().get_image()
"""
    ns = {'CAMERA': scenes.BENCH_CAMERA.tolist()}
    exec(script, ns)
    img = ns['img']
    assert img.shape == (128, 128, 4) and np.all(img[..., 3] == 1.0) and np.isfinite(img).all()
    assert 0.05 < img[..., :3].mean() < 2.0


def test_mid_size_scene_with_environment_vs_oracle(fresh, oracle_mod):
    '''config-4-shaped case at a size the oracle finishes in seconds: ~7k-triangle displaced blob in
    the cornell walls, equirect environment texture as world light, gather kernel (scene > LDS)'''
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    scene = scenes.scene_c4(n_side=24)
    assert scene[1].shape[0] == 10 + 12 * 24 * 24
    world = ([1.0, 1.0, 1.0, 1.0], 0)
    ref = setup_oracle(oracle_mod, scene, 64, 64, world=world)
    ref.render(8)
    for mode, tol in (('strict', 1e-4), ('fast', 1e-3)):
        reset_all()
        eng = _engine(None, scene, 64, 64, mode=mode, world=world)
        eng.render(8)
        if mode == 'fast':
            ctx().call('mpt_flush')
            assert ctx().get_option('last_kernel') == 2
        assert_parity(FilmTable().get_image(), ref.get_image(), *bounds(mode), what=f'c4-small {mode}')
    reset_all()


@pytest.mark.parametrize('name', ['s34', 's978', 'c4-7k'])
def test_counted_work_is_the_oracles_integer_for_integer(fresh, oracle_mod, name):
    '''the WORK is integer data and must be the reference's, not only the film.  The strict build (the reference's
    un-culled traversal, lbvh.py:324-345, and its bounce loop, path.py:25-62) against the oracle on the same frames: samples, rays,
    shaded hits, Sobol draws and bounces EQUAL, counter for counter, and the box tests (= internal nodes popped) and triangle tests equal
    up to the measured handful of grazing rays (below) -- the direct proof that the kernel does the reference's work and not less.
    The production build walks another tree in another order (culled, sorted 4-wide steps), so its box and triangle counts are its
    own; what must still agree with the strict build is the PATH structure: samples exactly, shaded hits / draws / bounces and --
    with option skip_dark = 0 -- rays up to the measured handful of paths whose fast-math arithmetic flipped a decision.'''
    from helpers import setup_oracle
    from ptina_amd.common import ctx, reset_all
    if name == 'c4-7k':
        scene, world, nx, ny, spp = scenes.scene_c4(n_side=24), ([1.0, 1.0, 1.0, 1.0], 0), 64, 64, 4
    else:
        scene, world = scenes.get_scene(name), None
        nx, ny, spp = (64, 64, 8) if name == 's34' else (96, 80, 8)
    ref = setup_oracle(oracle_mod, scene, nx, ny, world=world)
    ref.reset_counters()
    ref.render(spp)
    want = ref.counters()
    got = {}
    for mode in ('strict', 'fast'):
        reset_all()
        eng = _engine(None, scene, nx, ny, mode=mode, world=world)
        c = ctx()
        c.set_option('count', 1)
        if mode == 'fast':
            c.set_option('skip_dark', 0)
        c.call('mpt_reset_counters')
        eng.render(spp)
        c.call('mpt_flush')
        got[mode] = c.counters()
    reset_all()
    st, fa = got['strict'], got['fast']
    from helpers import _report
    _report('counted work %s: oracle %s | strict %s | fast %s' % (name, want, {k: st[k] for k in ('samples', 'rays', 'n_box', 'n_tri', 'n_shade', 'n_draws', 'bounces')},
                                                                   {k: fa[k] for k in ('samples', 'rays', 'n_box', 'n_tri', 'n_shade', 'n_draws', 'bounces')}))
    assert want['samples'] == nx * ny * spp
    # the path structure: exact
    for k in ('samples', 'rays', 'n_shade', 'n_draws', 'bounces'):
        assert st[k] == want[k], (name, k, st[k], want[k])
    # the traversal's tests.  Every ray is the oracle's up to the last bit of its direction (the device's sinf / cosf against glibc's in
    # the bounce sampler), and a ray that grazes a box within that bit passes the slab test in one implementation only: a handful of
    # subtrees more or less, never another hit (n_shade above is exact).  Measured, box / triangle tests: s34 +28 of 4 694 803 / +16 of
    # 2 435 233 (6e-6, 7e-6); s978 0 of 18 611 489 / +2 of 3 924 633; c4-7k +2 of 6 133 762 / 0 of 998 438; the bound is 2e-5 of the count
    assert st['n_node'] == st['n_box']                     # one box test per internal node popped (lbvh.py:338)
    for mine, theirs in (('n_box', 'n_int'), ('n_tri', 'n_leaf')):
        assert abs(st[mine] - want[theirs]) <= 2e-5 * want[theirs], (name, mine, st[mine], theirs, want[theirs])
    # production build: the same paths (measured: rays, shaded hits, draws and bounces EQUAL to the strict build's on all three scenes;
    # bound: 1e-4 of the strict count, for a path that flips a lobe under fast-math)
    assert fa['samples'] == st['samples']
    for k in ('n_shade', 'n_draws', 'bounces', 'rays'):
        assert abs(fa[k] - st[k]) <= max(1e-4 * st[k], 8), (name, k, fa[k], st[k])
    assert fa['n_draws'] == 2 * fa['samples'] + 6 * fa['n_shade']          # path.py:48,58,87: two jitter draws, six per shaded hit
    assert fa['n_box'] < st['n_box'] and fa['n_tri'] <= st['n_tri']        # the culled, ordered traversal tests less, never more


def test_large_scene_fast_vs_strict(fresh):
    '''config 4 at full triangle count (99 382) and a 200k random-triangle soup: device-built LBVH,
    SAH re-partition (c4) / raw LBVH (soup > sah_max is not reached here, so force it), fast vs strict'''
    from helpers import assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.sampling.sobol import SobolSampler
    from ptina_amd.common import ctx, reset_all
    for scene, world, tree in ((scenes.scene_c4(), ([1.0, 1.0, 1.0, 1.0], 0), 1),
                               (scenes.scene_random_tris(200000, seed=3), None, 0)):
        imgs = {}
        for mode in ('strict', 'fast'):
            reset_all()
            from ptina_amd.things import init_things
            init_things()
            ctx().set_option('tree', tree)
            eng = _engine(None, scene, 96, 96, mode=mode, world=world)
            eng.render(8)
            imgs[mode] = FilmTable().get_image()
            assert np.isfinite(imgs[mode]).all() and np.all(imgs[mode][..., 3] == 1)
        # the 200k-triangle soup has nearly coincident triangles everywhere: the pixels whose closest hit differs
        # between the ordered and the reference traversal carry the RMSE (measured 0.15 % / 3.1e-3; c4: 0.07 % / 4e-4)
        assert_parity(imgs['fast'], imgs['strict'], FAST[0], FAST[1], 4.2e-3 if tree == 0 else 1.2e-3, what=f'large scene tree={tree}')   # measured 3.19e-3 (LBVH) / 9.03e-4 (SAH): 1.3 x
    reset_all()


def test_gltf_compat_materials_parity(fresh, oracle_mod):
    '''materials as the reference's glTF loader leaves them: only base colour / metallic /
    roughness set, the other nine Disney parameters at zero -- specular 0, ior 0 (eta = 1/0)'''
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.common import reset_all
    v, m, _, _ = scenes.scene_s978()
    mats = [scenes.gltf_compat_material((0.8, 0.8, 0.8), 0.0, 0.5), scenes.gltf_compat_material((0.8, 0.05, 0.05), 0.0, 0.5),
            scenes.gltf_compat_material((0.05, 0.8, 0.05), 0.0, 0.5), scenes.gltf_compat_material((0.8, 0.6, 0.2), 0.1, 0.3)]
    scene = (v, m, mats, [])
    ref = setup_oracle(oracle_mod, scene, 64, 64)
    ref.render(16)
    want = ref.get_image()
    assert np.isfinite(want).all()
    for mode, tol in (('strict', 1e-4), ('fast', 1e-3)):
        reset_all()
        eng = _engine(None, scene, 64, 64, mode=mode)
        eng.render(16)
        assert_parity(FilmTable().get_image(), want, *bounds(mode), what=f'gltf-compat {mode}')
    reset_all()


def test_other_cameras_and_aspect_parity(fresh, oracle_mod):
    '''Camera.set_perspective / generate (camera.py:19-39) beyond the benchmark matrix: an off-axis
    perspective view built with tools.matrix on a non-square film, a view from inside the box towards
    a side wall, and an orthographic projection (w = 1 everywhere, parallel rays)'''
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.common import reset_all
    from ptina_amd.tools.matrix import perspective, lookat, orthogonal
    scene = scenes.scene_s978()
    cams = {
        'offaxis 96x56': (perspective(fov=45, aspect=96 / 56, near=0.1, far=100) @
                          lookat(pos=(0.3, 1.6, 0.0), back=(2.5, 1.2, 5.0), up=(0.1, 1.0, 0.0)), 96, 56),
        'inside 64x64': (perspective(fov=80, aspect=1, near=0.05, far=50) @
                         lookat(pos=(-2.0, 2.0, 0.0), back=(2.9, 0.3, 1.0)), 64, 64),
        'ortho 72x64': (orthogonal(size=2.4, aspect=72 / 64, near=-10, far=10) @
                        lookat(pos=(0.0, 2.0, 0.0), back=(0.5, 0.4, 3.0)), 72, 64),
    }
    for what, (cam, nx, ny) in cams.items():
        ref = setup_oracle(oracle_mod, scene, nx, ny, camera=cam)
        ref.render(8)
        want = ref.get_image()
        assert np.isfinite(want).all() and want[..., :3].max() > 0.05, what
        for mode, tol in (('strict', 1e-4), ('fast', 1e-3)):
            reset_all()
            eng = _engine(None, scene, nx, ny, mode=mode, camera=cam)
            eng.render(8)
            assert_parity(FilmTable().get_image(), want, *bounds(mode), what=f'{what} {mode}')
    reset_all()


LOBE_MATERIALS = {
    'glass': dict(basecolor=(0.9, 0.95, 1.0), roughness=0.08, transmission=0.9, ior=1.5, specular=0.5),
    'rough_glass': dict(basecolor=(0.8, 0.9, 0.8), roughness=0.45, transmission=0.6, ior=1.33, metallic=0.1),
    'clearcoat': dict(basecolor=(0.7, 0.1, 0.1), roughness=0.5, clearcoat=1.0, clearcoatGloss=0.9),
    'coat_on_metal': dict(basecolor=(0.9, 0.7, 0.3), roughness=0.3, metallic=0.9, clearcoat=0.5, clearcoatGloss=0.2),
    'cloth': dict(basecolor=(0.3, 0.2, 0.7), roughness=0.9, sheen=1.0, sheenTint=0.8, subsurface=0.7, specular=0.1),
    'tinted_spec': dict(basecolor=(0.1, 0.6, 0.2), roughness=0.2, specular=1.0, specularTint=1.0),
    'mirror': dict(basecolor=(0.95, 0.95, 0.95), roughness=0.0, metallic=1.0),
    'black': dict(basecolor=(0.0, 0.0, 0.0), roughness=0.5),
}


@pytest.mark.parametrize('name', sorted(LOBE_MATERIALS))
def test_disney_lobes_parity(fresh, oracle_mod, name):
    '''every branch of Disney.brdf / Disney.bounce (disney.py:53-233): transmission + refraction, the
    clearcoat lobe (whose GTR1 sample is NaN in the reference for alpha < 1 -- the path must die the
    same way), sheen / subsurface, tinted specular, a perfect mirror (alpha clamped to 0.001) and a
    black body, on both boxes of the 34-triangle scene.

    The two transmission materials first on the same scene with the boxes raised off the floor, at the ordinary bounds (round 5:
    the lobes themselves are as well-conditioned as any other).  On the scene as it stands the boxes' bottom faces COINCIDE with
    the floor quad, a ray inside a glass box meets both at the same depth, and which one the reference keeps hangs on the last
    bit of its depth formula and on its test order (strict `<`, lbvh.py:331): its f32 and f64 evaluations (the oracle's two
    builds) already disagree on 5-11 % of the pixels at 16 spp.  Rounds 1-4 read that as chaos of the lobe sampling; it is the
    tie (test_transmission_converged_shows_no_bias).  There the bound is calibrated against that spread instead of a fixed
    tolerance.'''
    from helpers import setup_oracle, assert_parity, image_stats, tile_means, _report
    from ptina_amd.things import FilmTable
    from ptina_amd.common import reset_all
    if LOBE_MATERIALS[name].get('transmission', 0.0) > 0.0:
        raised = _two_box_scene(LOBE_MATERIALS[name], 0.01)
        ref = setup_oracle(oracle_mod, raised, 64, 64)
        ref.render(16)
        for mode in ('strict', 'fast'):
            reset_all()
            eng = _engine(None, raised, 64, 64, mode=mode)
            eng.render(16)
            assert_parity(FilmTable().get_image(), ref.get_image(), *bounds(mode), what=f'{name} (boxes raised off the floor) {mode}')
        reset_all()
    v, m, mats, _ = scenes.scene_s34()
    mats = list(mats)
    mats[3] = scenes.material(**LOBE_MATERIALS[name])
    mats[4] = scenes.material(**LOBE_MATERIALS[name])
    scene = (v, m, mats, [])
    ref = setup_oracle(oracle_mod, scene, 64, 64)
    ref.render(16)
    want = ref.get_image()
    assert np.isfinite(want).all()
    chaotic = LOBE_MATERIALS[name].get('transmission', 0.0) > 0.0
    if chaotic:
        ref64 = setup_oracle(oracle_mod, scene, 64, 64, f64=True)
        ref64.render(16)
        d, refn, spread_rmse = image_stats(ref64.get_image(), want)
        spread_out = float((d > 1e-3 * (1 + refn)).mean())
        assert spread_out > 0.02, 'calibration: f32 and f64 oracle unexpectedly agree'
    for mode, tol in (('strict', 1e-4), ('fast', 1e-3)):
        reset_all()
        eng = _engine(None, scene, 64, 64, mode=mode)
        eng.render(16)
        got = FilmTable().get_image()
        assert np.isfinite(got).all()
        if not chaotic:
            assert_parity(got, want, *bounds(mode), what=f'{name} {mode}')
            continue
        d, refn, rel = image_stats(got, want)
        out = float((d > tol * (1 + refn)).mean())
        mean_err = abs(float(got[..., :3].mean()) / float(want[..., :3].mean()) - 1.0)
        print(f'{name} {mode}: outliers {out:.3%} rel-RMSE {rel:.2e} mean error {mean_err:.2%} '
              f'(f64-vs-f32 oracle: {spread_out:.3%}, {spread_rmse:.2e})')
        assert mean_err < 0.01, f'{name} {mode}: mean radiance off by {mean_err:.2%}'
        # a biased lobe cannot hide in the chaos: the 8x8-pixel tile means must agree as well as the
        # reference algorithm agrees with itself at another precision (plus 1 % of the mean radiance)
        tm_got, tm_want = tile_means(got), tile_means(want)
        scale = float(want[..., :3].mean())
        tile_err = float(np.sqrt(((tm_got - tm_want) ** 2).sum(axis=-1)).mean()) / scale
        tile_spread = float(np.sqrt(((tile_means(ref64.get_image()) - tm_want) ** 2).sum(axis=-1)).mean()) / scale
        print(f'{name} {mode}: mean 8x8-tile error {tile_err:.3%} of the mean radiance (f64-vs-f32 oracle: {tile_spread:.3%})')
        _report(f'{name} {mode}: tile error {tile_err:.3%} spread {tile_spread:.3%} outliers {out:.3%} relrmse {rel:.2e} mean_err {mean_err:.3%}')
        assert tile_err < (0.01 if mode == 'strict' else 0.01 + 2 * tile_spread), \
            f'{name} {mode}: 8x8 tile means off by {tile_err:.2%} (spread {tile_spread:.2%})'
        if mode == 'strict':       # same arithmetic as the f32 oracle up to libm: only a few flips
            assert out < 0.03 and rel < 1e-2, f'{name} strict: {out:.3%} outliers, rel-RMSE {rel:.2e}'
        else:                      # within twice the spread the reference algorithm shows itself
            assert out < 2 * spread_out + 0.02 and rel < 2 * spread_rmse + 1e-2, \
                f'{name} fast: {out:.3%} outliers (spread {spread_out:.3%}), rel-RMSE {rel:.2e} (spread {spread_rmse:.2e})'
    reset_all()


def _two_box_scene(material, lift):
    '''the 34-triangle scene with both boxes of `material`, standing on the floor as in scene_s34 (lift = 0: their bottom faces
    COINCIDE with the floor quad) or raised by `lift`'''
    parts = [scenes.cornell_walls(), scenes.box((-0.7, 1.2 + lift, -0.6), (0.6, 1.2, 0.6), 18.0, 3),
             scenes.box((0.75, 0.6 + lift, 0.55), (0.6, 0.6, 0.6), -17.0, 4)]
    v, m = scenes._compose(parts)
    return v, m, list(scenes.WALL_MATERIALS) + [scenes.material(**material), scenes.material(**material)], []


@pytest.mark.parametrize('name,lift', [('glass', 0.01), ('rough_glass', 0.01), ('glass', 0.0), ('rough_glass', 0.0)])
def test_transmission_converged_shows_no_bias(fresh, oracle_mod, name, lift):
    '''VERDICT r04 next #6.  At 16 spp the transmission lobes (disney.py:67-72,160-200; microfacet.py:13-27) can only be bounded by
    the spread of the reference's own algorithm (test_disney_lobes_parity).  Converged, a flipped path is one sample in thousands and a
    BIAS -- a lobe weighted wrongly, a term dropped -- would stay while the noise falls.  32 x 32 pixels at 512, 2048 and 8192 spp,
    the same Sobol points in every run, both builds against the f64 oracle with the f32 oracle beside them.'''
    from helpers import setup_oracle, tile_means, _report
    from ptina_amd.things import FilmTable
    from ptina_amd.common import reset_all
    nx = ny = 32
    stops = (512, 2048, 8192)
    scene = _two_box_scene(LOBE_MATERIALS[name], lift)
    imgs = {}
    for key, f64 in (('o64', True), ('o32', False)):
        ref = setup_oracle(oracle_mod, scene, nx, ny, f64=f64)
        done = 0
        for n in stops:
            ref.render(n - done)
            done = n
            imgs[(key, n)] = ref.get_image()
    for mode in ('strict', 'fast'):
        reset_all()
        eng = _engine(None, scene, nx, ny, mode=mode)
        done = 0
        for n in stops:
            eng.render(n - done)
            done = n
            imgs[(mode, n)] = FilmTable().get_image().copy()
            assert np.isfinite(imgs[(mode, n)]).all() and np.all(FilmTable().get_raw()[:, 3] == n)
    reset_all()
    tm = {k: tile_means(im) for k, im in imgs.items()}

    def rms(a, b, n):                  # tile means of a against b at n spp: rms over the tiles of the rgb distance, in units of the mean radiance
        scale = float(imgs[(b, n)][..., :3].mean())
        return float(np.sqrt(((tm[(a, n)] - tm[(b, n)]) ** 2).sum(axis=-1).mean())) / scale

    def mean_err(a, b, n):
        return abs(float(imgs[(a, n)][..., :3].mean()) / float(imgs[(b, n)][..., :3].mean()) - 1.0)

    def worst(a, b, n):
        scale = float(imgs[(b, n)][..., :3].mean())
        return float(np.sqrt(((tm[(a, n)] - tm[(b, n)]) ** 2).sum(axis=-1)).max()) / scale

    for mode in ('strict', 'fast'):
        msg = f'{name} lift {lift} converged {mode}: ' + '; '.join(
            f'{n} spp: mean vs f64 {mean_err(mode, "o64", n):.4%} (f32 oracle {mean_err("o32", "o64", n):.4%}), tile rms vs f64 {rms(mode, "o64", n):.4%} '
            f'worst {worst(mode, "o64", n):.4%} (f32 oracle: sigma {rms("o32", "o64", n):.4%} worst {worst("o32", "o64", n):.4%}), vs f32 oracle {rms(mode, "o32", n):.4%}'
            for n in stops)
        print(msg)
        _report(msg)
    last = stops[-1]
    sigma = rms('o32', 'o64', last)
    if lift > 0.0:
        # No coincident surfaces: nothing but the BSDF, the sampler and the traversal.  Measured (MI355X): strict build = the f32 oracle to
        # 1e-6 of the mean radiance; production build mean radiance within 0.0044 % at every sample count (bound 0.1 %), tile rms 0.031 %
        # at 512 spp falling to 0.0085 % at 8192 (noise: 1 / sqrt(16) = 0.25 x; a bias would stay) = 0.96 sigma (glass) / 1.0 sigma (rough glass)
        for mode in ('strict', 'fast'):
            for n in stops:
                assert mean_err(mode, 'o64', n) < 1e-3, (name, mode, n, mean_err(mode, 'o64', n))
                assert rms(mode, 'o64', n) < 4.1e-4, (name, mode, n, rms(mode, 'o64', n))              # 1.3 x the largest measured (0.0313 %)
            assert rms(mode, 'o64', last) <= 1.3 * sigma, (name, mode, rms(mode, 'o64', last), sigma)
            # worst tile: the f32 oracle's own is 2.84 sigma (sigma is an rms over 16 tiles), the production build's 3.48 sigma on glass
            assert worst(mode, 'o64', last) <= 4.5 * sigma, (name, mode, worst(mode, 'o64', last), sigma)
    else:
        # The boxes stand ON the floor: their bottom faces coincide with the floor quad, and a ray inside a glass box meets both at the
        # same depth.  Which of the two the reference keeps (strict `<`, lbvh.py:331), and whether the next ray -- which starts at
        # o + depth d, exactly on the shared plane, a hair above it or a hair below -- meets the coincident partner again (r > 0,
        # geometries.py:132), hang on the LAST BIT of every operation in between: its own f32 and f64 evaluations already disagree
        # there (THAT is the "chaos" of the transmission materials; with the boxes raised it is gone: above) and converge to each
        # other only slowly.  The strict build does the reference's operations in the reference's order and converges WITH the f32
        # oracle.  The production build's last bits fall differently (fused multiply-adds, v_rcp_f32; in the oracle a fused hitpos
        # alone moves the mean radiance by 0.4 %, a reciprocal one ulp off by 0.5 %: profiles/r05_ab_experiments.json), and its film of
        # such a scene settles elsewhere: a documented deviation on coincident transmissive geometry, bounded here at 1.3 x the
        # measurement (glass: mean radiance 0.28 %, tile rms 1.75 %; rough glass 0.094 %, 0.40 %), not a bias of the BSDF (above).
        # Scenes that need the reference's film there have the strict build (option "mode").
        assert mean_err('strict', 'o64', last) < 1e-3
        assert rms('strict', 'o64', last) <= 1.3 * sigma, (name, rms('strict', 'o64', last), sigma)
        mb, rb = (3.7e-3, 2.3e-2) if name == 'glass' else (1.3e-3, 5.2e-3)
        for n in stops:
            assert mean_err('fast', 'o64', n) < mb and rms('fast', 'o64', n) < rb, (name, n, mean_err('fast', 'o64', n), rms('fast', 'o64', n))


def test_many_lights(fresh, oracle_mod):
    '''a full light pool (64 lights, alternating POINT / AREA): first-hit-in-index-order semantics of
    LightPool.hit and the samp.z light pick of LightPool._sample'''
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.tools.matrix import translate
    from ptina_amd.common import reset_all
    rng = np.random.default_rng(11)
    rot = np.eye(4)
    rot[:3, :3] = [[1, 0, 0], [0, 0, 1], [0, -1, 0]]
    lights = []
    for k in range(64):
        pos = rng.uniform([-1.6, 2.2, -1.6], [1.6, 3.8, 1.6])
        col = rng.uniform(0.5, 3.0, 3)
        if k % 2:
            lights.append((translate(pos) @ rot, col * 4, 0.15, 'AREA'))
        else:
            lights.append((translate(pos), col, 0.1, 'POINT'))
    scene = scenes.scene_s34()
    ref = setup_oracle(oracle_mod, scene, 48, 48, lights=lights)
    ref.render(16)
    for mode, tol in (('strict', 1e-4), ('fast', 1e-3)):
        reset_all()
        eng = _engine(None, scene, 48, 48, mode=mode, lights=lights)
        eng.render(16)
        assert_parity(FilmTable().get_image(), ref.get_image(), *bounds(mode), what=f'64 lights {mode}')
    reset_all()


# ---------------------------------------------------------------- round 2: ordering, BASELINE configs 3-5 as stated
def test_state_change_between_partial_batches_is_ordered(fresh):
    '''frames enqueued below the batch size, then a call that enqueues work on the main stream
    (Sobol reset, preview pass, counter reset), then more frames: the pipelined launch on the aux /
    render streams must see that work.  One-frame launches (batch = 1) are the reference order.'''
    from ptina_amd.things import FilmTable
    from ptina_amd.sampling.sobol import SobolSampler
    from ptina_amd.engine.preview import PreviewEngine
    from ptina_amd.common import ctx, reset_all
    films = {}
    for batch in (1, 32):
        reset_all()
        eng = _engine(None, scenes.scene_s978(), 128, 96, mode='fast')
        ctx().set_option('batch', batch)
        for rep in range(3):                       # several rounds so that stale events would be picked up
            eng.render(5)
            SobolSampler().reset()
            eng.render(32)
            eng.render(3)
            PreviewEngine().render()
            eng.render(7)
            ctx().call('mpt_reset_counters')
            eng.render(2)
            ctx().call('mpt_sobol_update', 3)
            eng.render(30)
        films[batch] = (FilmTable().get_raw(0), FilmTable().get_raw(1), FilmTable().get_raw(2))
    reset_all()
    for a, b in zip(films[1], films[32]):
        assert np.array_equal(a, b)
    assert np.all(films[1][0][:, 3] == 3 * (5 + 32 + 3 + 7 + 2 + 30))


C5_SEED_REF_CAN_BUILD = 12346


def test_device_sah_pass_above_a_million_faces_and_its_host_fallback(fresh):
    '''round-3 ADVICE: (high) 2.6 M random triangles -- a level of the device SAH pass there has up to ~79 000 segments at 32
    bins, three times what the round-3 workspace held -- build on the device without falling back: every triangle in exactly
    one slot of the collapse, every wide node but the root with one parent, every pixel counted, and the film within the FAST
    bounds of the film through the host pass's tree; (medium) a device pass that fails AFTER it wrote the node records (test
    door sah_inject_fail) is not an error any more: the host pass re-packs the records and the film is the host tree's bit for bit'''
    from helpers import assert_parity
    from ptina_amd.things import FilmTable, BVHTree
    from ptina_amd.common import ctx, reset_all
    n = 2_600_000
    scene = scenes.scene_random_tris(n, seed=7)
    nx, ny, spp = 96, 96, 2
    films = {}
    for key, opts in (('dev', {'sah_build': 1}), ('host', {'sah_build': 0}), ('fallback', {'sah_build': 1, 'sah_inject_fail': 1})):
        reset_all()
        eng = _engine(None, scene, nx, ny, mode='fast', max_faces=n + 1)
        c = ctx()
        for k, v in opts.items():
            c.set_option(k, v)
        BVHTree().build()
        assert c.get_option('sah_fallback') == (1 if key == 'fallback' else 0), key
        if key == 'dev':
            w, q = _wide_records(c, n)
            ids = w[:, 6, :].view(np.int32)
            nw = c.get_option('wide_nodes')
            assert nw > 0 and 2 < c.get_option('fast_depth') <= 62
            assert np.array_equal(np.sort(ids[ids > 0]), np.arange(1, nw))
            assert np.array_equal(np.sort(~ids[(ids < 0) & (ids != ~n)]), np.arange(n))
            del w, q, ids
        eng.render(spp)
        raw = FilmTable().get_raw().reshape(nx, ny, 4)
        assert np.all(raw[..., 3] == spp) and np.isfinite(raw).all()
        films[key] = FilmTable().get_image().copy()
    reset_all()
    assert np.array_equal(films['fallback'], films['host'])
    assert_parity(films['dev'], films['host'], *FAST, what='2.6 M triangles: device SAH tree vs host SAH tree')   # measured: identical


def test_config5_reference_build_fails_on_the_stated_scene(fresh, oracle_mod):
    '''BASELINE configs[4] as generated (seed 12345): three of the 1 M centroids share a 30-bit Morton
    code, and the reference's own hierarchy (tree/lbvh.py:93-146, no index tie-break) is corrupted by
    it -- its genAABBs raises 'AABB step never stop' (lbvh.py:251-261).  The oracle restates that and
    fails the same way; the product (64-bit code<<32|index keys) must build and render it anyway.'''
    from ptina_amd.things import init_things, ModelPool, BVHTree, FilmTable
    from ptina_amd.common import ctx, reset_all
    n = 1_000_000
    scene = scenes.scene_random_tris(n)            # seed 12345, BASELINE's scene
    o = oracle_mod.Oracle(threads=1)
    o.load_model(scene[0], scene[1])
    with pytest.raises(RuntimeError, match='AABB step never stop'):
        o.build_tree()
    del o
    init_things()
    ModelPool().load(scene[0], scene[1])
    trees = {}
    for gpu in (1, 0):
        ctx().set_option('gpu_build', gpu)
        BVHTree().build()
        trees[gpu] = BVHTree().to_numpy()
    for k in ('mc', 'leaf', 'child', 'bmin', 'bmax', 'depth'):
        assert np.array_equal(trees[0][k], trees[1][k]), k
    assert np.array_equal(np.sort(trees[1]['leaf']), np.arange(n))
    reset_all()
    nx = ny = 1024
    spp = 16
    eng = _engine(None, scene, nx, ny, mode='fast')
    eng.render(spp)
    raw = FilmTable().get_raw().reshape(nx, ny, 4)
    assert np.all(raw[..., 3] == spp) and np.isfinite(raw).all() and raw[..., :3].min() >= 0
    fast = FilmTable().get_image()
    x0, x1 = 508, 516
    reset_all()
    eng = _engine(None, scene, nx, ny, mode='strict', slab=(x0, x1))
    eng.render(spp)
    strict = FilmTable().get_image()
    reset_all()
    from helpers import assert_parity
    assert_parity(fast[x0:x1], strict[x0:x1], FAST[0], FAST[1], 2.3e-3, what='C5 (seed 12345) fast vs strict, 8 columns')   # measured 1.74e-3 (bound = 1.3 x)


def test_config5_one_million_triangles_vs_oracle(fresh, oracle_mod):
    '''BASELINE configs[4] at full size on the first seed whose Morton codes the reference's build can
    handle: device-built LBVH == host build == the oracle's tree node for node; 1024x1024x16 fast render
    with every pixel counted; an 8-column window of both builds against the oracle'''
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import init_things, ModelPool, BVHTree, FilmTable
    from ptina_amd.common import ctx, reset_all
    n = 1_000_000
    scene = scenes.scene_random_tris(n, seed=C5_SEED_REF_CAN_BUILD)
    nx = ny = 1024
    spp = 16
    x0, x1 = 508, 516
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    ref = setup_oracle(oracle_mod, scene, nx, ny, threads=min(cores, 16))
    want_tree = ref.get_tree(n)
    init_things()
    ModelPool().load(scene[0], scene[1])
    for gpu in (1, 0):
        ctx().set_option('gpu_build', gpu)
        BVHTree().build()
        t = BVHTree().to_numpy()
        for k in ('mc', 'leaf', 'child', 'bmin', 'bmax'):
            assert np.array_equal(t[k], want_tree[k]), (gpu, k)
    reset_all()
    ref.set_window(x0, x1)
    ref.render(spp)
    want = ref.get_image()[x0:x1]
    eng = _engine(None, scene, nx, ny, mode='fast')
    assert ctx().get_option('gpu_build') == 1
    eng.render(spp)
    raw = FilmTable().get_raw().reshape(nx, ny, 4)
    assert np.all(raw[..., 3] == spp) and np.isfinite(raw).all() and raw[..., :3].min() >= 0
    # 8192 pixels at 16 spp over a triangle soup: a handful of pixels whose closest hit flips between two
    # nearly coincident triangles carry the RMSE (measured: 0.10 % outliers, rel-RMSE 2.9e-3, max diff 0.14)
    assert_parity(FilmTable().get_image()[x0:x1], want, FAST[0], FAST[1], 3.8e-3, what='C5 1M triangles fast, 8 columns x 16 spp')   # measured 2.91e-3 (bound = 1.3 x)
    assert ctx().get_option('last_kernel') == 2           # gather kernel over the 4-wide collapse (465 k nodes, 8-bit child boxes)
    # the same with the exact child boxes (option wide_quant = 0: 128-B records)
    ctx().set_option('wide_quant', 0)
    FilmTable().clear()
    ctx().call('mpt_sobol_reset', 64)
    eng.render(spp)
    raw = FilmTable().get_raw().reshape(nx, ny, 4)
    assert ctx().get_option('last_kernel') == 2 and np.all(raw[..., 3] == spp) and np.isfinite(raw).all()
    assert_parity(FilmTable().get_image()[x0:x1], want, FAST[0], FAST[1], 3.8e-3, what='C5 1M triangles fast, exact 4-wide boxes, 8 columns x 16 spp')   # measured 2.91e-3
    # the same through the binary tree (option wide = 0)
    ctx().set_option('wide', 0)
    FilmTable().clear()
    ctx().call('mpt_sobol_reset', 64)
    eng.render(spp)
    raw = FilmTable().get_raw().reshape(nx, ny, 4)
    assert ctx().get_option('last_kernel') == 0 and np.all(raw[..., 3] == spp) and np.isfinite(raw).all()
    assert_parity(FilmTable().get_image()[x0:x1], want, FAST[0], FAST[1], 3.8e-3, what='C5 1M triangles fast over the binary tree, 8 columns x 16 spp')   # measured 2.91e-3
    reset_all()
    eng = _engine(None, scene, nx, ny, mode='strict', slab=(x0, x1))
    eng.render(spp)
    # one pixel of the 8192 picks the other of two nearly coincident triangles (libm's last bit): 0.012 %
    # outliers, but its 0.03 difference alone puts the window's rel-RMSE at 3.4e-4 (measured)
    assert_parity(FilmTable().get_image()[x0:x1], want, STRICT[0], STRICT[1], 4.5e-4, what='C5 1M triangles strict, 8 columns x 16 spp')   # measured 3.42e-4 (bound = 1.3 x)
    reset_all()


def test_config4_full_scene_with_environment_vs_oracle(fresh, oracle_mod):
    '''BASELINE configs[3] as stated: 99 382 triangles in the cornell walls, equirect environment image
    as world light (light/world.py:22-29), 1024x1024; 8 spp over the whole film (every pixel counted),
    a 32-column window of both builds against the oracle'''
    from helpers import setup_oracle, assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    scene = scenes.scene_c4()
    assert scene[1].shape[0] == 99382
    world = ([1.0, 1.0, 1.0, 1.0], 0)
    nx = ny = 1024
    spp = 8
    x0, x1 = 496, 528
    ref = setup_oracle(oracle_mod, scene, nx, ny, world=world)
    ref.set_window(x0, x1)
    ref.render(spp)
    want = ref.get_image()[x0:x1]
    eng = _engine(None, scene, nx, ny, mode='fast', world=world)
    eng.render(spp)
    ctx().call('mpt_flush')
    assert ctx().get_option('last_kernel') == 2           # gather kernel over 4-wide nodes: the scene does not fit LDS
    raw = FilmTable().get_raw().reshape(nx, ny, 4)
    assert np.all(raw[..., 3] == spp) and np.isfinite(raw).all() and raw[..., :3].min() >= 0
    # one pixel next to the environment map's sun lobe differs by 0.79 at 8 spp: it alone is 1.4e-3 of rel-RMSE
    assert_parity(FilmTable().get_image()[x0:x1], want, FAST[0], FAST[1], 2.0e-3, what='C4 99k triangles + env fast, 32 columns x 8 spp')   # measured 1.51e-3 (bound = 1.3 x)
    reset_all()
    eng = _engine(None, scene, nx, ny, mode='strict', world=world, slab=(x0, x1))
    eng.render(spp)
    assert_parity(FilmTable().get_image()[x0:x1], want, *STRICT, what='C4 99k triangles + env strict, 32 columns x 8 spp')
    reset_all()


def test_headline_film_is_the_same_over_4wide_and_binary_nodes(fresh):
    '''BASELINE configs[1] at full size (512 x 512 x 32 spp) through the LDS-resident kernel over the 4-wide nodes (shipped) and over
    the binary nodes: another tree shape, another order among equally distant candidates, the same closest hits -- measured
    bit-identical in every pixel; the bound leaves room for a tie or two'''
    from helpers import assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    films = {}
    for lds_wide in (1, 0):
        reset_all()
        eng = _engine(None, scenes.scene_s978(), 512, 512, mode='fast')
        ctx().set_option('lds_wide', lds_wide)
        eng.render(32)
        films[lds_wide] = (FilmTable().get_raw().copy(), FilmTable().get_image().copy())
        assert ctx().get_option('last_kernel') == (5 if lds_wide else 1)
    reset_all()
    same = (films[1][0].view(np.uint32) == films[0][0].view(np.uint32)).all(axis=-1).mean()
    assert np.all(films[1][0][:, 3] == 32) and same >= 0.9999, same
    assert_parity(films[1][1], films[0][1], *FAST, what='headline film: 4-wide vs binary nodes in LDS')


@pytest.mark.gpu
def test_scaled_ray_distances_do_not_show(fresh):
    '''render_kernel_lds4 holds 1/d, o/d and tbest multiplied by a power of two while a ray is traversed (MptRenderParams::t_scale,
    from the scene's bounding sphere and the camera's near plane: DESIGN 3.1 step 33) so that an FMA's clamp bit stands for
    max(t, 0).  The scale must not show: a camera 2 000 units away behind a 0.2 degree lens, a camera inside the box, and the
    scene shrunk / blown up by 1 024 (a power of two: the same bits up to the exponent) give the film of the LDS-resident
    kernel over the binary nodes, which keeps distances unscaled -- bit for bit up to a tie or two'''
    from helpers import assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    from ptina_amd.tools.matrix import perspective, lookat, translate
    base = scenes.scene_s978()

    def scaled(k):
        v = base[0].copy()
        v[..., :3] *= k                                                      # positions; normals and texcoords stay
        return (v,) + tuple(base[1:])
    far_dir = np.array([0.25, 0.2, 1.0]); far_dir /= np.linalg.norm(far_dir)
    target = np.array([0.0, 1.6, 0.0])
    cases = {
        'far tele': (base, perspective(fov=0.2, aspect=1, near=1.0, far=5000) @
                     lookat(pos=tuple(target), back=tuple(target + 2000.0 * far_dir)), None),
        'inside': (base, perspective(fov=80, aspect=1, near=0.05, far=50) @ lookat(pos=(-2.0, 2.0, 0.0), back=(2.9, 0.3, 1.0)), None),
        'scene / 1024': (scaled(1.0 / 1024), None, 1.0 / 1024),
        'scene x 1024': (scaled(1024.0), None, 1024.0),
    }
    for what, (scene, cam, k) in cases.items():
        if cam is None:                                                       # the benchmark view, moved with the scene
            from ptina_amd.tools.matrix import scale as _scale_matrix
            cam = scenes.BENCH_CAMERA @ _scale_matrix(1.0 / k)
        films = {}
        for lds_wide in (1, 0):
            reset_all()
            # (the default light -- POINT at (1, 2, 3), radius 0.5, colour 32 -- moved and dimmed with the scene: the same irradiance)
            lights = None if k is None else [(translate([1.0 * k, 2.0 * k, 3.0 * k]), np.array([32.0, 32.0, 32.0]) * k * k, 0.5 * k, 'POINT')]
            eng = _engine(None, scene, 96, 96, mode='fast', camera=cam, lights=lights)
            ctx().set_option('lds_wide', lds_wide)
            eng.render(8)
            films[lds_wide] = (FilmTable().get_raw().copy(), FilmTable().get_image().copy())
            assert ctx().get_option('last_kernel') == (5 if lds_wide else 1), what
        same = (films[1][0].view(np.uint32) == films[0][0].view(np.uint32)).all(axis=-1).mean()
        assert np.isfinite(films[1][1]).all() and films[1][1][..., :3].max() > 0.01, what
        assert same >= 0.999, (what, same)
        assert_parity(films[1][1], films[0][1], *FAST, what='scaled ray distances: ' + what)
    reset_all()


def test_kernel_ladder_by_scene_size(fresh):
    '''which kernel serves a scene is decided by what fits a CU's 160 KiB of LDS beside the stacks (miptina.cpp): the 4-wide
    nodes with exact boxes (112 B a node, about half a node per triangle) and the triangles, else the 4-wide 8-bit nodes gathered
    from L2; the LDS-resident kernel over the binary nodes (72 B, one per triangle: it fits less) is what option lds_wide = 0
    asks for.  The benchmark model and the same with a finer sphere: the same picture from each kernel as from the gather kernel
    that can serve them all (equal up to ties: FAST bounds)'''
    from helpers import assert_parity
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    from ptina_amd.scenes import cornell_walls, bumpy_sphere, _compose
    base = scenes.scene_s978()
    for segments, rings, lds_wide, want_kernel in ((22, 23, 1, 5), (22, 23, 0, 1), (28, 27, 1, 2), (28, 27, 0, 2)):
        vertices, mtlids = _compose([cornell_walls(), bumpy_sphere(segments=segments, rings=rings)])
        scene = (vertices, mtlids, base[2], [])
        n = mtlids.shape[0]
        films = {}
        for lds in (1, 0):
            reset_all()
            eng = _engine(None, scene, 96, 80, mode='fast')
            c = ctx()
            c.set_option('lds', lds)
            c.set_option('lds_wide', lds_wide)
            eng.render(8)
            films[lds] = FilmTable().get_image().copy()
            raw = FilmTable().get_raw()
            assert np.all(raw[:, 3] == 8) and np.isfinite(raw).all()
            assert c.get_option('last_kernel') == (want_kernel if lds else 2), (n, lds, c.get_option('last_kernel'), c.get_option('wide_nodes'), c.get_option('wide_stack'))
        assert_parity(films[1], films[0], *FAST, what='%d triangles: kernel %d vs the gather kernel' % (n, want_kernel))
    reset_all()


def _same_film(a, b, what):
    '''bit equality of two films (columns, rows, 4), saying where they differ when they do'''
    d = (np.ascontiguousarray(a).view(np.uint32) != np.ascontiguousarray(b).view(np.uint32)).any(axis=-1)
    if not d.any():
        return True
    cx, cy = np.nonzero(d)
    pytest.fail('%s differs in %d pixels: columns %d..%d, rows %d..%d, first (%d, %d) %s vs %s' % (
        what, len(cx), cx.min(), cx.max(), cy.min(), cy.max(), cx[0], cy[0], a[cx[0], cy[0]], b[cx[0], cy[0]]))


def test_config3_eight_stripe_shares_reassemble_bit_identically(fresh):
    '''BASELINE configs[2] AS STATED -- 2048x2048 at 256 spp (the tests ran it at 2 and 32) -- split as
    bench.py --gpus 8 splits it: stripes of 16 columns dealt to 8 ranks (mpt_set_stripes(16, r, 8)); render(256) is eight
    pipelined launches of 32 frames (the first finalises its own tiles, the others keep the combine pass); all eight shares
    rendered one after the other on this GPU must reassemble into exactly the single-GPU film, every pixel counted 256 times'''
    from ptina_amd.things import FilmTable
    from ptina_amd.common import ctx, reset_all
    from ptina_amd.dist import stripe_columns
    from ptina_amd import _lib
    n, spp, R = 2048, 256, 8
    scene = scenes.scene_s978()
    eng = _engine(None, scene, n, n, mode='fast', max_filmsize=n * n)
    eng.render(spp)
    full = FilmTable().get_raw().reshape(n, n, 4).copy()
    assert np.all(full[..., 3] == spp) and np.isfinite(full).all()
    tiled = np.zeros_like(full)
    for r in range(R):
        reset_all()
        eng = _engine(None, scene, n, n, mode='fast', max_filmsize=n * n)
        ctx().call('mpt_set_stripes', 16, r, R)
        eng.render(spp)
        part = FilmTable().get_raw().reshape(n, n, 4)
        cols = stripe_columns(n, R, r)
        assert np.all(part[np.setdiff1d(np.arange(n), cols)] == 0)
        if not np.array_equal(part[cols].view(np.uint32), full[cols].view(np.uint32)):
            # which of the two is off?  (diagnosis only: render the whole film once more)
            mine = part[cols].copy()
            reset_all()
            eng = _engine(None, scene, n, n, mode='fast', max_filmsize=n * n)
            eng.render(spp)
            again = FilmTable().get_raw().reshape(n, n, 4).copy()
            same_full = np.array_equal(again.view(np.uint32), full.view(np.uint32))
            same_part = np.array_equal(again[cols].view(np.uint32), mine.view(np.uint32))
            d = (full.view(np.uint32) != again.view(np.uint32)).any(axis=-1)
            cx, cy = np.nonzero(d)
            where = 'whole films differ in %d pixels, columns %s rows %s' % (len(cx), sorted(set((cx // 8 * 8).tolist()))[:12], sorted(set((cy // 8 * 8).tolist()))[:12]) if len(cx) else ''
            _same_film(mine, full[cols], 'share %d as rendered (a second whole-film render equals the first: %s, equals the share: %s; %s)' % (r, same_full, same_part, where))
        if r == 0:
            tiled[cols] = part[cols]                      # the root's own share is already in its film
        else:
            # the gather's device half for this peer: its 16 stripes packed into ONE message by copy_pieces,
            # the message copied, and scattered into the root's film (mpt_comm_selftest; RCCL itself needs 2 GPUs)
            flat = np.ascontiguousarray(tiled.reshape(-1, 4))
            ctx().call('mpt_comm_selftest', r, R, 0, _lib.fptr(np.ascontiguousarray(part.reshape(-1, 4))), _lib.fptr(flat))
            tiled = flat.reshape(n, n, 4)
    reset_all()
    assert _same_film(tiled, full, 'the reassembled film')


@pytest.mark.parametrize('world', [2, 3, 8])
def test_gather_pack_and_scatter_kernels_follow_the_plan(fresh, world):
    '''the device half of mpt_comm_gather_film on one GPU (mpt_comm_selftest: pack by the sender's plan, one
    message, scatter on the root) for R in {2, 3, 8}, a ragged film, striped and slab splits and a root that is not
    rank 0: every peer's columns arrive, nothing else of the root's film is touched'''
    from ptina_amd import _lib
    from ptina_amd.common import ctx
    from ptina_amd.dist import comm_plan
    from ptina_amd.things import init_things, FilmTable
    init_things()
    rng = np.random.default_rng(world)
    nx, ny = 102, 37
    FilmTable().set_size(nx, ny)
    truth = rng.normal(size=(nx * ny, 4)).astype(np.float32)
    for stripe in (16, 0, 32):
        if stripe:
            ctx().call('mpt_set_stripes', stripe, 0, world)
        else:
            ctx().call('mpt_set_slab', 0, nx)            # back to slab mode (stripe_w = 0)
        for root in (0, world - 1):
            plans = [comm_plan(nx, ny, stripe, r, world) for r in range(world)]
            out = np.full((nx * ny, 4), np.float32(-3.0))
            for o, n in plans[root]:
                out[o:o + n] = truth[o:o + n]
            for r in range(world):
                if r == root:
                    continue
                share = np.full((nx * ny, 4), np.float32(-9.0 - r))          # garbage outside the share
                for o, n in plans[r]:
                    share[o:o + n] = truth[o:o + n]
                before = out.copy()
                ctx().call('mpt_comm_selftest', r, world, root, _lib.fptr(share), _lib.fptr(out))
                touched = np.zeros(nx * ny, bool)
                for o, n in plans[r]:
                    touched[o:o + n] = True
                assert np.array_equal(out[~touched], before[~touched]), (stripe, root, r)
            assert np.array_equal(out, truth), (stripe, root)


def test_bench_refuses_more_gpus_than_the_box_has(fresh):
    '''bench.py --gpus 2 on a one-GPU box: the launcher starts two ranks, rank 1 finds no device for its
    LOCAL_RANK and the whole run exits non-zero (no silent pile-up on GPU 0, no n_gpus: 1 line)'''
    import subprocess
    import sys
    from ptina_amd import _lib
    if _lib.load_library().mpt_device_count() >= 2:
        pytest.skip('needs a box with exactly one GPU')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    env['MIPTINA_LAUNCH_TIMEOUT'] = '240'
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert '"n_gpus"' not in p.stdout


@pytest.mark.parametrize('world', [2, 4])
def test_processes_render_their_stripes_concurrently_on_one_gpu(fresh, tmp_path, world):
    '''a dress rehearsal of the N-GPU run with real HIP contexts.  `bench.py --gpus R --host-gather` starts R
    fresh processes through bench.py's own launcher (ptina_amd.dist.launch_ranks: self-launch before any GPU call, a private
    rendezvous directory, RANK / WORLD_SIZE / MASTER_*), every rank creates its own context on the one GPU, takes its stripes
    (mpt_set_stripes(16, r, R)) and renders them WHILE the others render theirs; warm-up, counting steps, timed steps and the
    2048 x 2048 leg run as in a multi-GPU run, with the PhaseLog watching.  RCCL refuses two ranks on one device, so the shares
    travel through files (dist.HostFilm) instead of ncclSend / ncclRecv -- the only part of an R > 1 run this does not execute.
    The film rank 0 assembles must be the single-process film of the same steps, bit for bit.  (R = 4, not 8: a GPU box
    allows six processes on its card, and this test runner is one of them.)'''
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from ptina_amd.common import reset_all
    reset_all()                                          # (this process keeps its HIP context but holds no film / slabs meanwhile)
    common = ['--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-pmc', '--no-configs']
    env = dict(os.environ, MIPTINA_PHASE_TIMEOUT='240', MIPTINA_LAUNCH_TIMEOUT='500')
    ref = tmp_path / 'ref.npy'
    one = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--save-film', str(ref)] + common,
                         capture_output=True, text=True, env=env, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    got = tmp_path / f'r{world}.npy'
    run = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(world), '--host-gather', '--c3-steps', '1',
                          '--c3-spp', '32', '--save-film', str(got)] + common, capture_output=True, text=True, env=env, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == world and 'REHEARSAL' in line['config']['parallelism'] and line['value'] > 0
    assert line['c3']['n_gpus'] == world and line['c3']['msamples_s'] > 0
    a, b = np.load(ref), np.load(got)
    assert a.shape == b.shape == (512 * 512, 4)
    assert np.all(a[:, 3] == a[0, 3]) and a[0, 3] > 0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    print(f'{world} processes on one GPU: {line["value"]:.0f} Msamples/s (a rehearsal, not a measurement), film == the single-process film, '
          f'{int(a[0, 3])} samples per pixel')
