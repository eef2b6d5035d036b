'''
ctypes binding of libmiptina.so (include/miptina.h) -- the only door between PTina's
Python object API and the gfx950 kernels.  There is NO CPU fallback: if the library is
missing or no MI355X is visible, every entry point raises.
'''

import ctypes as C
import os
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# MIPTINA_LIB: another build of the same library (A/B runs of compiler flags, tools/gpu_round.sh)
LIB_PATH = os.environ.get('MIPTINA_LIB') or os.path.join(HERE, 'libmiptina.so')

MODE_FAST, MODE_STRICT = 0, 1
LIGHT_TYPES = {'POINT': 1, 'AREA': 2}          # LightPool.TYPES, light/__init__.py:11


# kinds of mpt_unit_eval (include/miptina.h, enum MPT_UNIT_*): name -> (kind, input columns, output columns)
UNIT_KINDS = {
    'schlick': (0, 1, 1), 'dielectric': (1, 3, 1), 'gtr1': (2, 2, 1), 'gtr2': (3, 2, 1), 'smithggx': (4, 2, 1),
    'sample_gtr1': (5, 3, 3), 'sample_gtr2': (6, 3, 3), 'tanspace': (7, 6, 3), 'spherical': (8, 2, 3),
    'dir2tex': (9, 3, 2), 'reflect': (10, 6, 3), 'refract': (11, 7, 4), 'box': (12, 12, 3), 'face': (13, 30, 9),
    'sphere': (14, 10, 1), 'area': (15, 15, 4), 'disney_brdf': (16, 24, 3), 'disney_bounce': (17, 24, 7),
    'power_heuristic': (18, 2, 1), 'wanghash': (19, 1, 1), 'wanghash2': (20, 2, 1),
}


class Caps(C.Structure):
    _fields_ = [(k, C.c_int32) for k in
                ('max_faces', 'max_texels', 'max_materials', 'max_textures', 'max_lights',
                 'max_filmsize', 'max_filmpasses')]


class Counters(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in
                ('samples', 'rays', 'n_box', 'n_tri', 'n_shade', 'n_draws', 'bounces', 'n_node',
                 'it_node', 'it_leaf', 'it_shade', 'it_new',
                 'pl_local', 'pl_batches', 'pl_batch_lanes', 'pl_prim', 'pl_tidle', 'pl_sidle', 'pl_trips', 'pl_taken')]

    def asdict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


# every symbol include/miptina.h declares: name -> (restype, argtypes)
_vp, _i, _fp, _ip = C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int32)
SIGNATURES = {
    'mpt_last_error': (C.c_char_p, []),
    'mpt_device_count': (_i, []),
    'mpt_version': (_i, []),
    'mpt_create': (_vp, [C.POINTER(Caps), _i]),
    'mpt_destroy': (None, [_vp]),
    'mpt_set_option': (_i, [_vp, C.c_char_p, _i]),
    'mpt_get_option': (_i, [_vp, C.c_char_p, C.POINTER(_i)]),
    'mpt_set_size': (_i, [_vp, _i, _i]),
    'mpt_get_size': (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    'mpt_set_slab': (_i, [_vp, _i, _i]),
    'mpt_set_stripes': (_i, [_vp, _i, _i, _i]),
    'mpt_load_model': (_i, [_vp, _fp, _ip, _i]),
    'mpt_load_materials': (_i, [_vp, _fp, _ip, _i]),
    'mpt_reset_images': (_i, [_vp]),
    'mpt_load_image': (_i, [_vp, _fp, _i, _i, C.POINTER(_i)]),
    'mpt_build_tree': (_i, [_vp]),
    'mpt_get_tree': (_i, [_vp, _ip, _ip, _fp, _fp, _ip, _ip]),
    'mpt_get_wide': (_i, [_vp, _fp, _fp, _i, C.POINTER(_i)]),
    'mpt_sah_workspace': (_i, [_i, C.c_int64, C.POINTER(C.c_int64)]),
    'mpt_get_oct8': (_i, [_vp, _fp, _ip, _i, C.POINTER(_i)]),
    'mpt_set_camera': (_i, [_vp, _fp, _fp]),
    'mpt_clear_lights': (_i, [_vp]),
    'mpt_add_light': (_i, [_vp, _i, _fp, _fp, _fp, C.c_float, C.POINTER(_i)]),
    'mpt_set_world_light': (_i, [_vp, _fp, _i]),
    'mpt_sobol_init': (_i, [_vp, _ip, _i, _i]),
    'mpt_sobol_reset': (_i, [_vp, _i]),
    'mpt_sobol_update': (_i, [_vp, _i]),
    'mpt_sobol_get': (_i, [_vp, _ip, _fp, _ip]),
    'mpt_render': (_i, [_vp, _i]),
    'mpt_render_preview': (_i, [_vp, _i]),
    'mpt_flush': (_i, [_vp]),
    'mpt_synchronize': (_i, [_vp]),
    'mpt_clear': (_i, [_vp, _i]),
    'mpt_get_image': (_i, [_vp, _i, _fp]),
    'mpt_hint_image': (_i, [_vp, _i, _fp]),
    'mpt_fast_export_image': (_i, [_vp, _i, _fp]),
    'mpt_get_film_raw': (_i, [_vp, _i, _fp]),
    'mpt_resolve': (_i, [_vp, _i]),
    'mpt_host_alloc': (_vp, [C.c_size_t]),
    'mpt_host_free': (None, [_vp]),
    'mpt_get_counters': (_i, [_vp, C.POINTER(Counters)]),
    'mpt_get_timeline': (_i, [_vp, C.POINTER(C.c_ulonglong), _i, C.POINTER(_i)]),
    'mpt_reset_counters': (_i, [_vp]),
    'mpt_get_lane_hist': (_i, [_vp, C.POINTER(C.c_uint64), _i]),
    'mpt_probe_kernel': (_i, [_vp, _i, _i, C.POINTER(C.c_double)]),
    'mpt_stress_copies': (_i, [_vp, _i, _i]),
    'mpt_kernel_time': (_i, [_vp, C.POINTER(C.c_double), C.POINTER(_i)]),
    'mpt_unit_eval': (_i, [_vp, _i, _vp, _i, _vp, _i, _i]),
    'mpt_comm_unique_id': (_i, [C.c_char_p]),
    'mpt_comm_init': (_i, [_vp, C.c_char_p, _i, _i]),
    'mpt_comm_plan': (_i, [_i, _i, _i, _i, _i, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _i]),
    'mpt_comm_gather_film': (_i, [_vp, _i, _i]),
    'mpt_comm_selftest': (_i, [_vp, _i, _i, _i, _fp, _fp]),
    'mpt_comm_barrier': (_i, [_vp]),
    'mpt_comm_allreduce_max': (_i, [_vp, C.POINTER(C.c_double)]),
    'mpt_comm_destroy': (_i, [_vp]),
}

_lib = None
_lock = threading.Lock()


def load_library():
    '''dlopen libmiptina.so and bind every declared symbol (no GPU needed for this)'''
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f'{LIB_PATH} not found: build it with `make -C ptina_amd/csrc` '
                    '(or __graft_entry__.build()); there is no CPU fallback')
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                f = getattr(lib, name)
                f.restype = res
                f.argtypes = args
            _lib = lib
    return _lib


def last_error():
    return load_library().mpt_last_error().decode('utf-8', 'replace')


def check(rc):
    if rc != 0:
        raise RuntimeError(last_error())


class HostBuffer:
    '''a page-locked host buffer of libmiptina; goes back to the pool when the last numpy view dies'''
    _pool = {}                                  # bytes -> [addresses]
    _pool_cap = 4

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        free = HostBuffer._pool.get(self.nbytes)
        if free:
            self.addr = free.pop()
        else:
            self.addr = load_library().mpt_host_alloc(self.nbytes)
            if not self.addr:
                raise MemoryError(last_error())

    def __del__(self):
        try:
            free = HostBuffer._pool.setdefault(self.nbytes, [])
            if len(free) < HostBuffer._pool_cap:
                free.append(self.addr)
            else:
                load_library().mpt_host_free(self.addr)
        except Exception:
            pass


def host_array(shape, dtype=np.float32):
    '''a fresh numpy array on page-locked memory (read-backs into it are a single DMA); the buffer is
    recycled as soon as the array and every view of it are gone (plain reference counting: the array's
    base is a ctypes buffer that owns the HostBuffer, and nothing points back)'''
    count = int(np.prod(shape))
    hb = HostBuffer(max(count * np.dtype(dtype).itemsize, 1))
    buf = (C.c_char * hb.nbytes).from_address(hb.addr)
    buf._owner = hb
    return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)


def fptr(a):
    return a.ctypes.data_as(_fp)


def iptr(a):
    return a.ctypes.data_as(_ip)


def devices_isolated():
    '''True when the launcher shows each rank its own GPU(s) only (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES set per task, SLURM --gpus-per-task): LOCAL_RANK then does not index the visible devices'''
    return any(os.environ.get(k) for k in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'))


def rank_device(ndev):
    '''the device of this rank: MIPTINA_DEVICE if given; LOCAL_RANK when the node's GPUs are all visible (a rank
    whose LOCAL_RANK has no device then fails in mpt_create -- "device out of range" -- instead of piling onto
    somebody else's GPU); device 0 when the launcher isolated ONE GPU per rank'''
    if os.environ.get('MIPTINA_DEVICE'):
        return int(os.environ['MIPTINA_DEVICE'])
    lr = int(os.environ.get('LOCAL_RANK', '0'))
    if ndev == 1 and lr > 0 and devices_isolated():
        return 0
    return lr


class Context:
    '''one device context = the reference's set of singletons (things.py:20-28)'''

    def __init__(self, device=None, **caps):
        lib = load_library()
        c = Caps(max_faces=2**21, max_texels=2**22, max_materials=2**6, max_textures=2**6,
                 max_lights=2**6, max_filmsize=2**21, max_filmpasses=3)
        for k, v in caps.items():
            setattr(c, k, int(v))
        self.caps = c
        if device is None:
            device = rank_device(lib.mpt_device_count())
        self.device = device
        self.lib = lib
        h = lib.mpt_create(C.byref(c), device)
        if not h:
            raise RuntimeError(last_error())
        self.h = C.c_void_p(h)

    def close(self):
        if getattr(self, 'h', None):
            self.lib.mpt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call(self, name, *args):
        check(getattr(self.lib, name)(self.h, *args))

    def set_option(self, key, value):
        self.call('mpt_set_option', key.encode(), int(value))

    def get_option(self, key):
        v = C.c_int(0)
        self.call('mpt_get_option', key.encode(), C.byref(v))
        return v.value

    def unit_eval(self, name, rows):
        '''test door (mpt_unit_eval): one device function of the hot path on rows of inputs, by the build the
        context's mode selects; rows f32 (i32 for the hash kinds), returns [n, out_cols] of the same type'''
        kind, nin, nout = UNIT_KINDS[name]
        dt = np.int32 if name.startswith('wanghash') else np.float32
        a = np.ascontiguousarray(np.asarray(rows, dt).reshape(-1, nin))
        out = np.zeros((a.shape[0], nout), dt)
        self.call('mpt_unit_eval', kind, a.ctypes.data_as(C.c_void_p), nin, out.ctypes.data_as(C.c_void_p), nout, a.shape[0])
        return out

    def counters(self):
        cnt = Counters()
        self.call('mpt_get_counters', C.byref(cnt))
        return cnt.asdict()

    def kernel_time(self):
        ms, n = C.c_double(0), C.c_int(0)
        self.call('mpt_kernel_time', C.byref(ms), C.byref(n))
        return ms.value, n.value


_ctx = None


def get_context(**caps):
    '''the process-wide context; created on first use (init_things passes capacities)'''
    global _ctx
    if _ctx is None:
        _ctx = Context(**caps)
    return _ctx


def have_context():
    return _ctx is not None


def drop_context():
    global _ctx
    if _ctx is not None:
        _ctx.close()
        _ctx = None
