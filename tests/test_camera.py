"""pt_camera constructors of the C ABI: the three forms the reference's host side holds a camera
in (uniform-block matrices, src/camera.h's Camera struct, the live application's look-at call)
give one and the same pt_camera.  Host code only: no GPU."""
import math

import numpy as np
import pytest


def _cols(cam):
    return np.array([list(cam.c0), list(cam.c1), list(cam.c2)], dtype=np.float64)


def _quat_from_matrix(r):
    """Unit quaternion (x, y, z, w) of a rotation matrix with columns right, up, back."""
    m = np.asarray(r, dtype=np.float64)
    w = math.sqrt(max(0.0, 1.0 + m[0, 0] + m[1, 1] + m[2, 2])) / 2
    x = (m[2, 1] - m[1, 2]) / (4 * w)
    y = (m[0, 2] - m[2, 0]) / (4 * w)
    z = (m[1, 0] - m[0, 1]) / (4 * w)
    return np.array([x, y, z, w])


def test_look_at_matches_the_binding(mipt):
    eye, target = (3.0, 2.0, -7.5), (0.5, 1.0, 0.25)
    fovy, aspect = math.radians(47.0), 16 / 9
    cam = mipt.camera_look_at(eye, target, None, fovy, aspect)
    ref = mipt.look_at_camera(eye, target, fovy_deg=47.0, aspect=aspect)
    np.testing.assert_allclose(_cols(cam), _cols(ref), rtol=0, atol=1e-6)
    np.testing.assert_allclose(list(cam.eye), eye, rtol=1e-7)
    assert abs(cam.sx - ref.sx) < 1e-6 and abs(cam.sy - ref.sy) < 1e-6
    assert cam.mode == mipt.PT_CAMERA_PINHOLE
    r = _cols(cam)                                   # rows: right, up, back -- an orthonormal, right-handed frame
    np.testing.assert_allclose(r @ r.T, np.eye(3), atol=1e-6)
    assert np.linalg.det(r) > 0.999
    back = np.array(eye) - np.array(target)
    np.testing.assert_allclose(r[2], back / np.linalg.norm(back), atol=1e-6)


def test_live_application_camera(mipt):
    """look_at((0,0,-10) -> origin), src/main_granite.cpp:306-308: looks along +z."""
    cam = mipt.camera_look_at((0, 0, -10), (0, 0, 0), (0, 1, 0), math.radians(60), 1280 / 720,
                              mode=mipt.PT_CAMERA_REFERENCE)
    np.testing.assert_allclose(list(cam.c2), [0, 0, -1], atol=1e-7)     # back = -forward
    np.testing.assert_allclose(list(cam.c1), [0, 1, 0], atol=1e-7)
    np.testing.assert_allclose(list(cam.c0), [-1, 0, 0], atol=1e-7)
    assert cam.mode == mipt.PT_CAMERA_REFERENCE


def test_three_forms_agree(mipt):
    eye, target = (-4.0, 6.0, 9.0), (1.0, 0.0, -2.0)
    fovy, w, h = math.radians(70.0), 960, 720                     # the legacy window, src/main.cpp:843
    a = mipt.camera_look_at(eye, target, None, fovy, w / h)
    rot = _cols(a).T                                              # columns right, up, back
    b = mipt.camera_from_orientation(_quat_from_matrix(rot), eye, fovy, (w, h))
    np.testing.assert_allclose(_cols(b), _cols(a), atol=2e-6)
    assert abs(b.sx - a.sx) < 1e-6 and abs(b.sy - a.sy) < 1e-6
    # Camera::getForwardVector = orientation * (0,0,-1), src/camera.h:20
    np.testing.assert_allclose(-np.array(list(b.c2)), (np.array(target) - eye) / np.linalg.norm(np.array(target) - eye), atol=2e-6)
    # the uniform block's matrices: inv_view = [R | eye], inv_projection = diag(aspect tan, tan, ., .)
    inv_view = np.eye(4)
    inv_view[:3, :3] = rot
    inv_view[:3, 3] = eye
    t = math.tan(fovy / 2)
    inv_proj = np.diag([t * w / h, t, 1.0, 1.0])
    c = mipt.camera_from_matrices(inv_view.T.reshape(-1), inv_proj.T.reshape(-1))     # column-major
    np.testing.assert_allclose(_cols(c), _cols(a), atol=1e-6)
    np.testing.assert_allclose(list(c.eye), eye, rtol=1e-7)
    assert abs(c.sx - a.sx) < 1e-6 and abs(c.sy - a.sy) < 1e-6
    d = mipt.camera_from_matrices(inv_view.T.reshape(-1), inv_proj.T.reshape(-1), camera_position=(1, 2, 3))
    assert list(d.eye) == [1.0, 2.0, 3.0]


def test_rejects_what_is_no_camera(mipt):
    with pytest.raises(mipt.MiptError):
        mipt.camera_look_at((1, 1, 1), (1, 1, 1), None, 1.0, 1.0)            # eye == target
    with pytest.raises(mipt.MiptError):
        mipt.camera_look_at((0, 0, 0), (0, 5, 0), (0, 1, 0), 1.0, 1.0)        # up parallel to the view
    with pytest.raises(mipt.MiptError):
        mipt.camera_look_at((0, 0, 0), (0, 0, 1), None, 0.0, 1.0)             # no field of view
    with pytest.raises(mipt.MiptError):
        mipt.camera_from_orientation((0, 0, 0, 0), (0, 0, 0), 1.0, (4, 4))    # null quaternion
    with pytest.raises(mipt.MiptError):
        mipt.camera_from_orientation((0, 0, 0, 1), (0, 0, 0), 1.0, (0, 4))    # empty viewport
    with pytest.raises(mipt.MiptError):
        mipt.camera_from_matrices([float("nan")] * 16, [1.0] * 16)
    with pytest.raises(mipt.MiptError):
        mipt.camera_look_at((0, 0, 0), (0, 0, 1), None, 1.0, 1.0, mode=7)
