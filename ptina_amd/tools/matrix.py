'''
4x4 numpy matrix helpers with the reference's names and conventions (tools/matrix.py):
column vectors, right-handed, camera looks down -z, OpenGL clip space.
'''

import numpy as np


def identity():
    return np.eye(4)


def affine(lin, pos):
    m = np.eye(4)
    m[:3, :3] = np.asarray(lin, float)
    m[:3, 3] = np.asarray(pos, float)
    return m


def lookat(pos=(0, 0, 0), back=(0, 0, 3), up=(0, 1, 1e-12)):
    '''world -> view for an eye at pos + back looking at pos'''
    pos = np.asarray(pos, float)
    back = np.asarray(back, float)
    fwd = -back / np.linalg.norm(back)
    right = np.cross(fwd, np.asarray(up, float))
    right /= np.linalg.norm(right)
    upv = np.cross(right, fwd)
    view2world = affine(np.stack([right, upv, -fwd], axis=1), pos + back)
    return np.linalg.inv(view2world)


def ortho(left=-1, right=1, bottom=-1, top=1, near=-100, far=100):
    m = np.eye(4)
    m[0, 0] = 2 / (right - left)
    m[1, 1] = 2 / (top - bottom)
    m[2, 2] = -2 / (far - near)
    m[:3, 3] = [-(right + left) / (right - left), -(top + bottom) / (top - bottom),
                -(far + near) / (far - near)]
    return m


def frustum(left=-1, right=1, bottom=-1, top=1, near=1, far=100):
    m = np.zeros((4, 4))
    m[0, 0] = 2 * near / (right - left)
    m[1, 1] = 2 * near / (top - bottom)
    m[0, 2] = (right + left) / (right - left)
    m[1, 2] = (top + bottom) / (top - bottom)
    m[2, 2] = -(far + near) / (far - near)
    m[2, 3] = -2 * far * near / (far - near)
    m[3, 2] = -1
    return m


def orthogonal(size=1, aspect=1, near=-100, far=100):
    return ortho(-size * aspect, size * aspect, -size, size, near, far)


def perspective(fov=60, aspect=1, near=0.05, far=500):
    t = np.tan(np.radians(fov) / 2)
    return frustum(-near * t * aspect, near * t * aspect, -near * t, near * t, near, far)


def scale(factor):
    return affine(np.diag(np.ones(3) * np.asarray(factor, float)), np.zeros(3))


def translate(offset):
    return affine(np.eye(3), np.ones(3) * np.asarray(offset, float))
