"""TEST INFRASTRUCTURE (the product never imports this).  The reference's camera arithmetic in binary32, step by step:
what a Rust run uploads as u_modelview / u_projection for a player at (pos, yaw, pitch).

  player transform   Decomposed { scale: 1, rot: Quaternion::from(Euler { x: pitch, y: yaw, z: 0 }), disp: pos }
                     (game/src/player.rs:124-131: at a reset pitch = Rad(1e-8), yaw = level.start_yaw(), pos = level.start_pos())
  camera             Decomposed { scale: 1, rot: identity, disp: (0, camera_height = 0.12, 0) }, a child of the player (player.rs:325-335)
  absolute           player.concat(camera)                                   (engine/src/transforms.rs:121)
  view               Matrix4::from(absolute.inverse_transform())              (engine/src/renderer.rs:78-87)
  projection         cgmath::perspective(Rad::from(Deg(65)), aspect * 1.2, 0.01, 100)  (player.rs:336-344, engine/src/projections.rs:93-101)

cgmath 0.18.0 (Cargo.lock:178) is a third-party crate that is not under /root/reference; its published formulas are restated:
  Quaternion::from(Euler)   the euclideanspace.net conversion on half angles
  q * v (rotate)            tmp = q.v x v + v * q.s;  (q.v x tmp) * 2 + v
  Quaternion * Quaternion   the Hamilton product, each component summed left to right
  magnitude2                s * s + v.magnitude2(), v.magnitude2() = dot(v, v) = (x*x + y*y) + z*z   <- the order round 4's C helper got wrong
  invert                    conjugate / magnitude2
  Decomposed::concat        rot = a.rot * b.rot;  disp = a.rot.rotate(b.disp * a.scale) + a.disp;  scale = a.scale * b.scale
  inverse_transform         s = 1 / scale;  r = rot.invert();  d = r.rotate(disp) * -s
  Matrix3::from(Quaternion) from the doubled components;  Matrix4::from(Decomposed) = (Matrix3 * scale), w column = disp
Every operation below is one numpy float32 operation (IEEE binary32, round to nearest even: what rustc emits for f32 without
fast-math); sin / cos / tan are glibc's sinf / cosf / tanf called through ctypes -- Rust's f32::sin lowers to the same libm entry
points on Linux, numpy's own float32 sin is a SIMD routine that may differ in the last place.
tests/test_pose_helpers.py holds rdoom_pose_from_player to this, bit for bit; tests/golden/make_golden.py generates pose 0 of every
level from it (the fixtures do not depend on the library they check)."""
import ctypes
import ctypes.util

import numpy as np

F = np.float32
_libm = ctypes.CDLL(ctypes.util.find_library('m') or 'libm.so.6')
for _n in ('sinf', 'cosf', 'tanf'):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]


def sinf(x):
    return F(_libm.sinf(float(F(x))))


def cosf(x):
    return F(_libm.cosf(float(F(x))))


def tanf(x):
    return F(_libm.tanf(float(F(x))))


def _cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def _rotate(q, v):  # impl Mul<Vector3> for Quaternion
    s, qv = q[0], q[1:]
    c = _cross(qv, v)
    tmp = (c[0] + v[0] * s, c[1] + v[1] * s, c[2] + v[2] * s)
    c2 = _cross(qv, tmp)
    two = F(2.0)
    return (c2[0] * two + v[0], c2[1] * two + v[1], c2[2] * two + v[2])


def _qmul(a, b):  # impl Mul for Quaternion
    return (a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
            a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3], a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1])


def pose_from_player(pos, yaw, pitch, width, height):
    """-> (modelview[16], projection[16]) column-major float32, as the reference uploads them"""
    with np.errstate(all='ignore'):
        pos = [F(x) for x in pos]
        yaw, pitch, half, zero, one = F(yaw), F(pitch), F(0.5), F(0.0), F(1.0)
        sx, cx, sy, cy, sz, cz = sinf(pitch * half), cosf(pitch * half), sinf(yaw * half), cosf(yaw * half), sinf(zero * half), cosf(zero * half)
        player = (-sx * sy * sz + cx * cy * cz, sx * cy * cz + sy * sz * cx, -sx * sz * cy + sy * cx * cz, sx * sy * cz + sz * cx * cy)
        identity = (one, zero, zero, zero)
        scale = one * one
        rot = _qmul(player, identity)
        cam = (zero * one, F(0.12) * one, zero * one)
        rc = _rotate(player, cam)
        disp = (rc[0] + pos[0], rc[1] + pos[1], rc[2] + pos[2])
        s = one / scale
        vv = (rot[1] * rot[1] + rot[2] * rot[2]) + rot[3] * rot[3]   # Vector3::dot: the element products summed x + y + z
        mag2 = rot[0] * rot[0] + vv                                  # Quaternion::magnitude2
        r = (rot[0] / mag2, -rot[1] / mag2, -rot[2] / mag2, -rot[3] / mag2)
        rd = _rotate(r, disp)
        d = (rd[0] * -s, rd[1] * -s, rd[2] * -s)
        x2, y2, z2 = r[1] + r[1], r[2] + r[2], r[3] + r[3]
        xx2, xy2, xz2, yy2, yz2, zz2 = x2 * r[1], x2 * r[2], x2 * r[3], y2 * r[2], y2 * r[3], z2 * r[3]
        sy2, sz2, sx2 = y2 * r[0], z2 * r[0], x2 * r[0]
        m3 = (one - yy2 - zz2, xy2 + sz2, xz2 - sy2, xy2 - sz2, one - xx2 - zz2, yz2 + sx2, xz2 + sy2, yz2 - sx2, one - xx2 - yy2)
        mv = np.zeros(16, np.float32)
        for c in range(3):
            for rr in range(3):
                mv[c * 4 + rr] = m3[c * 3 + rr] * s
        mv[12], mv[13], mv[14], mv[15] = d[0], d[1], d[2], one
        fovy = F(65.0) * F(np.pi / 180.0)
        near, far = F(0.01), F(100.0)
        aspect = (F(width) / F(height)) * F(1.2)
        f = one / tanf(fovy / F(2.0))
        pr = np.zeros(16, np.float32)
        pr[0], pr[5] = f / aspect, f
        pr[10] = (far + near) / (near - far)
        pr[11] = F(-1.0)
        pr[14] = (F(2.0) * far * near) / (near - far)
        return mv, pr
