'''
TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU oracle (oracle/ptina_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product package (ptina_amd) never does.

PARITY UNPINNED against real PTina output: Taichi cannot be installed in the build
container and the reference's tests hold no vectors for this path; see the header of
ptina_oracle.c.  The Sobol sampler is pinned independently against scipy.
'''

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.environ.get('ORACLE_BUILD_DIR') or os.path.join(HERE, '_build')    # _build_san: sanitizer build
JOE_KUO = os.path.join(os.path.dirname(HERE), 'ptina_amd', 'data', 'joe_kuo_21201.npz')

LIGHT_TYPES = {'POINT': 1, 'AREA': 2}


def build(force=False):
    '''compile the oracle with gcc (make -C oracle)'''
    so = os.path.join(BUILD, 'libptina_oracle.so')
    src = os.path.join(HERE, 'ptina_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', HERE, '-s'])
    return so


class Counters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in
                ('samples', 'rays', 'n_int', 'n_leaf', 'n_shade', 'n_draws', 'max_stack', 'bounces')]

    def asdict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


_libs = {}


def load(f64=False):
    key = bool(f64)
    if key in _libs:
        return _libs[key]
    build()
    name = 'libptina_oracle_f64.so' if f64 else 'libptina_oracle.so'
    lib = C.CDLL(os.path.join(BUILD, name))
    real = C.c_double if f64 else C.c_float
    rp = C.POINTER(real)
    fp = C.POINTER(C.c_float)
    ip = C.POINTER(C.c_int32)
    vp = C.c_void_p

    def sig(name, res, *args):
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = list(args)

    sig('orc_wanghash', C.c_int32, C.c_int32)
    sig('orc_wanghash2', C.c_int32, C.c_int32, C.c_int32)
    sig('orc_expand_bits', C.c_int32, C.c_int32)
    sig('orc_morton3d', C.c_int32, rp)
    sig('orc_clz', C.c_int32, C.c_int32)
    sig('orc_count_low_bits', C.c_int32, C.c_int32)
    sig('orc_construct_float', real, C.c_int32)
    sig('orc_box_intersect', C.c_int, rp, rp, rp, rp, rp, rp)
    sig('orc_face_intersect', C.c_int, rp, rp, rp, rp, rp, rp)
    sig('orc_sphere_intersect', real, rp, real, rp, rp)
    sig('orc_area_intersect', C.c_int, rp, rp, rp, rp, rp, rp, rp)
    sig('orc_disney_brdf', None, rp, rp, real, rp, rp, rp)
    sig('orc_disney_bounce', None, rp, rp, real, rp, rp, rp)
    sig('orc_power_heuristic', real, real, real)
    sig('orc_unit_microfacet', None, C.c_int, rp, rp)
    sig('orc_unit_common', None, C.c_int, rp, rp)
    sig('orc_unit_face_shading', None, rp, rp, real, real, rp, rp)
    sig('orc_sobol_vgrid', None, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
        C.POINTER(C.c_uint32), C.c_int, C.c_int, ip)
    sig('orc_create', vp)
    sig('orc_destroy', None, vp)
    sig('orc_set_threads', None, vp, C.c_int)
    sig('orc_set_size', None, vp, C.c_int, C.c_int)
    sig('orc_set_window', None, vp, C.c_int, C.c_int)
    sig('orc_set_stripes', None, vp, C.c_int, C.c_int, C.c_int)
    sig('orc_load_model', C.c_int, vp, fp, ip, C.c_int)
    sig('orc_load_materials', C.c_int, vp, fp, ip, C.c_int)
    sig('orc_reset_images', None, vp)
    sig('orc_add_image', C.c_int, vp, fp, C.c_int, C.c_int)
    sig('orc_build_tree', C.c_int, vp)
    sig('orc_get_tree', C.c_int, vp, ip, ip, fp, fp, ip)
    sig('orc_set_camera_v2w', None, vp, fp)
    sig('orc_clear_lights', None, vp)
    sig('orc_add_light', C.c_int, vp, C.c_int, fp, fp, fp, C.c_float)
    sig('orc_set_world', None, vp, fp, C.c_int)
    sig('orc_sobol_init', None, vp, ip, C.c_int, C.c_int)
    sig('orc_sobol_reset', None, vp, C.c_int)
    sig('orc_sobol_update', None, vp)
    sig('orc_sobol_get', C.c_int, vp, ip, rp)
    sig('orc_render', None, vp)
    sig('orc_render_preview', None, vp)
    sig('orc_clear', None, vp)
    sig('orc_get_image', None, vp, C.c_int, fp)
    sig('orc_fast_export_image', None, vp, C.c_int, fp)
    sig('orc_get_film_raw', None, vp, C.c_int, fp)
    sig('orc_get_film_real', None, vp, C.c_int, rp)
    sig('orc_get_counters', None, vp, C.POINTER(Counters))
    sig('orc_reset_counters', None, vp)
    sig('orc_trace_pixel', None, vp, C.c_int, C.c_int, rp)
    sig('orc_camera_generate', None, vp, real, real, rp, rp)
    sig('orc_intersect', C.c_int, vp, rp, rp, C.c_int, rp)
    lib._real = real
    lib._np_real = np.float64 if f64 else np.float32
    _libs[key] = lib
    return lib


def _ptr(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


_vgrid_cache = {}


def sobol_vgrid(nsamples=2**20, dim=21201, f64=False):
    '''direction-number grid by the oracle's C restatement of calc_sobol_vgrid'''
    key = (nsamples, dim)
    if key not in _vgrid_cache:
        lib = load(f64)
        z = np.load(JOE_KUO)
        s = np.ascontiguousarray(z['s'][:dim], np.uint8)
        a = np.ascontiguousarray(z['a'][:dim], np.uint32)
        m = np.ascontiguousarray(z['m'][:dim], np.uint32)
        L = int(np.ceil(np.log2(nsamples)))
        V = np.zeros((L + 1, dim), np.int32)
        lib.orc_sobol_vgrid(_ptr(s, C.c_uint8), _ptr(a, C.c_uint32), _ptr(m, C.c_uint32),
                            dim, L, _ptr(V, C.c_int32))
        _vgrid_cache[key] = V
    return _vgrid_cache[key]


def flatten_materials(materials):
    '''materials: list of lists of (fac, tex) pairs, as MaterialPool.load takes them
    (mtllib.py:58-77).  Unset parameters: fac 0, tex -1 (documented deviation Q6).'''
    m = len(materials)
    fac = np.zeros((max(m, 1), 12, 4), np.float32)
    tex = np.full((max(m, 1), 12), -1, np.int32)
    for i, mat in enumerate(materials):
        for k, (f, t) in enumerate(mat):
            if k >= 12:
                break
            if f is None:
                f = 1.0
            f = np.asarray(f, np.float64)
            if f.ndim == 0:
                f = np.full(4, float(f))
            elif f.shape[0] == 3:
                f = np.concatenate([f, [1.0]])
            fac[i, k] = f
            tex[i, k] = t
    return fac, tex


def usable_cpus():
    '''CPUs this process can really use at once: the affinity mask capped by the cgroup's CPU quota (a GPU box shows all 256 host
    cores but gives one GPU's share, cpu.max = 16 CPUs: 256 spinning OpenMP threads on 16 CPUs are 5 x SLOWER than 16)'''
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(-(-int(quota) // int(period)))))
    except (OSError, ValueError):
        pass
    return max(1, n)


class Oracle:
    '''one scene context of the CPU restatement, with PTina's call sequence'''

    def __init__(self, f64=False, threads=None, sobol=True):
        self.lib = load(f64)
        self.real = self.lib._real
        self.npreal = self.lib._np_real
        self.ctx = C.c_void_p(self.lib.orc_create())
        self.nx = self.ny = 0
        if threads is None:
            threads = usable_cpus()
        self.lib.orc_set_threads(self.ctx, int(threads))
        if sobol:
            V = sobol_vgrid()
            self.lib.orc_sobol_init(self.ctx, _ptr(V, C.c_int32), V.shape[0], V.shape[1])
            self.lib.orc_sobol_reset(self.ctx, 64)       # SobolSampler(skip=64), sobol.py:75

    def __del__(self):
        try:
            self.lib.orc_destroy(self.ctx)
        except Exception:
            pass

    # ---- scene ----
    def set_size(self, nx, ny):
        self.nx, self.ny = nx, ny
        self.lib.orc_set_size(self.ctx, nx, ny)

    def set_window(self, x0, x1):
        self.lib.orc_set_window(self.ctx, x0, x1)

    def set_stripes(self, width, index, modulo):
        self.lib.orc_set_stripes(self.ctx, width, index, modulo)

    def load_model(self, vertices, mtlids=None):
        v = np.ascontiguousarray(vertices, np.float32)
        n = v.shape[0] // 3
        m = None if mtlids is None else np.ascontiguousarray(mtlids, np.int32)
        self.lib.orc_load_model(self.ctx, _ptr(v, C.c_float),
                                None if m is None else _ptr(m, C.c_int32), n)

    def load_materials(self, materials):
        fac, tex = flatten_materials(materials)
        self.lib.orc_load_materials(self.ctx, _ptr(fac, C.c_float), _ptr(tex, C.c_int32),
                                    len(materials))

    def load_images(self, images):
        self.lib.orc_reset_images(self.ctx)
        for arr in images:
            arr = np.asarray(arr)
            if arr.dtype == np.uint8:
                arr = arr.astype(np.float32) / 255
            if arr.ndim == 2:
                arr = arr[:, :, None]
            if arr.shape[2] == 1:
                arr = np.repeat(arr, 3, axis=2)
            if arr.shape[2] == 3:
                arr = np.concatenate([arr, np.ones(arr.shape[:2] + (1,), arr.dtype)], axis=2)
            a = np.ascontiguousarray(arr, np.float32)
            self.lib.orc_add_image(self.ctx, _ptr(a, C.c_float), a.shape[0], a.shape[1])

    def build_tree(self):
        if self.lib.orc_build_tree(self.ctx) != 0:
            raise RuntimeError('AABB step never stop! hierarchy corrupted?')

    def get_tree(self, n):
        child = np.zeros((max(n - 1, 1), 2), np.int32)
        leaf = np.zeros(n, np.int32)
        bmin = np.zeros((max(n - 1, 1), 3), np.float32)
        bmax = np.zeros((max(n - 1, 1), 3), np.float32)
        mc = np.zeros(n, np.int32)
        self.lib.orc_get_tree(self.ctx, _ptr(child, C.c_int32), _ptr(leaf, C.c_int32),
                              _ptr(bmin, C.c_float), _ptr(bmax, C.c_float), _ptr(mc, C.c_int32))
        return dict(child=child[:n - 1], leaf=leaf, bmin=bmin[:n - 1], bmax=bmax[:n - 1], mc=mc)

    def set_camera(self, pers):
        '''Camera.set_perspective, camera.py:19-22: inverse in f64, stored f32'''
        v2w = np.ascontiguousarray(np.linalg.inv(np.asarray(pers, np.float64)), np.float32)
        self.lib.orc_set_camera_v2w(self.ctx, _ptr(v2w, C.c_float))

    def clear_lights(self):
        self.lib.orc_clear_lights(self.ctx)

    def add_light(self, world, color, size, type):
        '''LightPool.add, light/__init__.py:34-49'''
        world = np.asarray(world, np.float64)
        pos = world @ np.array([0, 0, 0, 1.0])
        pos = np.ascontiguousarray(pos[:3] / pos[3], np.float32)
        axes = np.ascontiguousarray(world[:3, :3], np.float32)
        color = np.ascontiguousarray(color, np.float32)
        return self.lib.orc_add_light(self.ctx, LIGHT_TYPES[type], _ptr(color, C.c_float),
                                      _ptr(pos, C.c_float), _ptr(axes, C.c_float), float(size))

    def set_world_light(self, fac, tex):
        f = np.ascontiguousarray(np.broadcast_to(np.asarray(fac, np.float32), (4,)))
        self.lib.orc_set_world(self.ctx, _ptr(f, C.c_float), int(tex))

    def load_scene(self, scene, camera=None):
        vertices, mtlids, materials, images = scene
        self.load_model(vertices, mtlids)
        self.load_materials(materials)
        self.load_images(images)
        self.build_tree()
        if camera is not None:
            self.set_camera(camera)

    # ---- sampler ----
    def sobol_reset(self, skip=64):
        self.lib.orc_sobol_reset(self.ctx, skip)

    def sobol_update(self):
        self.lib.orc_sobol_update(self.ctx)

    def sobol_state(self):
        D = 21201
        X = np.zeros(D, np.int32)
        P = np.zeros(D, self.npreal)
        t = self.lib.orc_sobol_get(self.ctx, _ptr(X, C.c_int32), _ptr(P, self.real))
        return t, X, P

    # ---- rendering ----
    def render(self, nframes=1):
        for _ in range(nframes):
            self.lib.orc_render(self.ctx)

    def render_preview(self):
        self.lib.orc_render_preview(self.ctx)

    def clear(self):
        self.lib.orc_clear(self.ctx)

    def get_image(self, id=0):
        out = np.empty((self.nx, self.ny, 4), np.float32)
        self.lib.orc_get_image(self.ctx, id, _ptr(out, C.c_float))
        return out

    def fast_export_image(self, out, id=0):
        self.lib.orc_fast_export_image(self.ctx, id, _ptr(out, C.c_float))

    def get_film_raw(self, id=0):
        out = np.empty((self.nx * self.ny, 4), np.float32)
        self.lib.orc_get_film_raw(self.ctx, id, _ptr(out, C.c_float))
        return out

    def get_film_real(self, id=0):
        '''raw sums in the build's own precision'''
        out = np.zeros((self.nx * self.ny, 4), self.npreal)
        self.lib.orc_get_film_real(self.ctx, id, _ptr(out, self.real))
        return out

    def counters(self):
        cnt = Counters()
        self.lib.orc_get_counters(self.ctx, C.byref(cnt))
        return cnt.asdict()

    def reset_counters(self):
        self.lib.orc_reset_counters(self.ctx)

    def trace_pixel(self, i, j):
        rgb = np.zeros(3, self.npreal)
        self.lib.orc_trace_pixel(self.ctx, i, j, _ptr(rgb, self.real))
        return rgb

    def camera_generate(self, x, y):
        o = np.zeros(3, self.npreal)
        d = np.zeros(3, self.npreal)
        self.lib.orc_camera_generate(self.ctx, x, y, _ptr(o, self.real), _ptr(d, self.real))
        return o, d

    def intersect(self, o, d, avoid=-1):
        o = np.ascontiguousarray(o, self.npreal)
        d = np.ascontiguousarray(d, self.npreal)
        out = np.zeros(4, self.npreal)
        hit = self.lib.orc_intersect(self.ctx, _ptr(o, self.real), _ptr(d, self.real), avoid,
                                     _ptr(out, self.real))
        return hit, float(out[0]), int(out[1]), float(out[2]), float(out[3])
