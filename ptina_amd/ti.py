'''
the handful of `ti.*` names PTina's driver scripts touch (exams/benchmark.py:7,37),
so that those scripts run against this package unchanged.  Nothing here compiles or
traces anything: kernels are prebuilt HIP code objects.
'''

cuda = 'hip:gfx950'
gpu = cuda
cpu = 'cpu'
opengl = 'opengl'
cc = 'cc'


def init(arch=None, **kwargs):
    if arch in (cpu, cc):
        raise RuntimeError('ptina_amd runs on MI355X only: there is no CPU backend')
    from . import _lib
    _lib.load_library()


def imshow(img, title='image'):
    import numpy as np
    a = np.asarray(img)
    print(f'[ptina_amd] imshow "{title}": shape {a.shape}, mean {float(a[..., :3].mean()):.6f}')


def imwrite(img, path):
    import numpy as np
    from PIL import Image
    a = np.asarray(img)[..., :3]
    a = (np.clip(np.swapaxes(a, 0, 1)[::-1], 0, 1) * 255).astype(np.uint8)
    Image.fromarray(a).save(path)
