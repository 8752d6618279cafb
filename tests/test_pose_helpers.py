"""rdoom_pose_from_player (the reference's own binary32 camera arithmetic: Decomposed / Quaternion of cgmath 0.18 as
game/src/player.rs:124-131, 325-345 + engine/src/renderer.rs:78-87 use them) against rdoom_pose_look (double precision,
rounded once) and against a float64 composition written here from the same formulas.
At a reset the two helpers agree to a few units in the last place; BASELINE config 2's pose (E1M1, spawn view) is the
from_player one -- in tests/test_gpu_raster_parity.py::test_static_320x200_single_pose and as pose 0 of every level in
tests/golden/poses.npy (oracle, HIP and GL readback fixtures are generated from it)."""
import numpy as np

import rust_doom_amd as rd
from util import GOLDEN, META_PATH, ensure_wad


def _ulps(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def _from_player_f64(pos, yaw, pitch):
    """the same chain in float64: Quaternion::from(Euler {x: pitch, y: yaw, z: 0}) -> concat with the camera at (0, 0.12, 0)
    -> inverse -> matrix (column-major 16)"""
    sx, cx, sy, cy = np.sin(pitch / 2), np.cos(pitch / 2), np.sin(yaw / 2), np.cos(yaw / 2)
    q = np.array([cx * cy, sx * cy, sy * cx, sx * sy])   # (s, x, y, z) with z = 0

    def rot(q, v):
        qv = q[1:]
        tmp = np.cross(qv, v) + v * q[0]
        return np.cross(qv, tmp) * 2 + v
    disp = rot(q, np.array([0, np.float64(np.float32(0.12)), 0])) + np.asarray(pos, np.float64)
    r = np.array([q[0], -q[1], -q[2], -q[3]]) / (q @ q)
    d = -rot(r, disp)
    s, x, y, z = r
    m3 = np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y + 2 * z * s, 2 * x * z - 2 * y * s],
                   [2 * x * y - 2 * z * s, 1 - 2 * x * x - 2 * z * z, 2 * y * z + 2 * x * s],
                   [2 * x * z + 2 * y * s, 2 * y * z - 2 * x * s, 1 - 2 * x * x - 2 * y * y]])   # columns
    out = np.zeros(16)
    for c in range(3):
        out[c * 4:c * 4 + 3] = m3[c]
    out[12:15], out[15] = d, 1
    return out


def test_spawn_pose_of_e1m1_two_helpers_agree_to_a_few_ulps():
    built = rd.Wad(ensure_wad(), META_PATH).build_level(0)
    pos, yaw = built.start()
    a = rd.pose_from_player(pos, yaw, 1e-8, 320, 200)
    b = rd.pose_look((pos[0], np.float32(pos[1]) + np.float32(0.12), pos[2]), yaw, 1e-8, 320, 200)
    diff = np.abs(a['modelview'] - b['modelview'])
    print('spawn pose: largest |from_player - look| = %.3g (%d ulps of the largest entry), projection %d ulps'
          % (diff.max(), int(_ulps(a['modelview'][12:15], b['modelview'][12:15]).max()), int(_ulps(a['projection'], b['projection']).max())))
    assert diff[:12].max() <= 2e-7                       # rotation part: unit-scale entries, an ulp or two
    assert _ulps(a['modelview'][12:15], b['modelview'][12:15]).max() <= 4   # translation: a few ulps of |eye| ~ 30
    assert _ulps(a['projection'], b['projection']).max() <= 2
    assert a['modelview'][15] == 1.0 and not a['modelview'][[3, 7, 11]].any()
    golden = np.load(GOLDEN + '/poses.npy')               # pose 0 of every level is the from_player pose
    assert np.array_equal(golden[0, 0, :16], a['modelview']) and np.array_equal(golden[0, 0, 16:32], a['projection'])


def test_from_player_equals_the_binary32_transcription_bit_for_bit():
    """oracle/camera.py restates the cgmath chain one numpy float32 operation at a time (sinf / cosf / tanf from glibc, as Rust's
    f32::sin lowers to); the C helper must produce the SAME BITS -- including Quaternion::magnitude2's summation order
    (s s + ((x x + y y) + z z)), which round 4's helper had as a left-to-right sum.  The golden poses are
    generated from the transcription, so this also ties rdoom_pose_from_player to tests/golden/poses.npy."""
    from oracle import camera
    rng = np.random.RandomState(20260922)
    order_matters = 0
    for k in range(6000):
        pos = rng.uniform(-60, 60, 3).astype(np.float32)
        yaw = np.float32(rng.uniform(-7, 7))
        pitch = np.float32(1e-8) if k % 3 == 0 else np.float32(rng.uniform(-1.5, 1.5))
        w, h = ((320, 200), (1920, 1080), (1366, 768))[k % 3]
        a = rd.pose_from_player(pos, yaw, pitch, w, h)
        mv, pr = camera.pose_from_player(pos, yaw, pitch, w, h)
        assert np.array_equal(a['modelview'].view(np.uint32), mv.view(np.uint32)), (k, pos, yaw, pitch)
        assert np.array_equal(a['projection'].view(np.uint32), pr.view(np.uint32)), (k, w, h)
        # how often the summation order decides a bit (so that the test above really discriminates)
        sx, cx, sy, cy = camera.sinf(pitch * np.float32(0.5)), camera.cosf(pitch * np.float32(0.5)), camera.sinf(yaw * np.float32(0.5)), camera.cosf(yaw * np.float32(0.5))
        q = (cx * cy, sx * cy, sy * cx, sx * sy)
        order_matters += int((q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]) != (q[0] * q[0] + ((q[1] * q[1] + q[2] * q[2]) + q[3] * q[3])))
    assert order_matters > 50, order_matters
    golden = np.load(GOLDEN + '/poses.npy')
    wad = rd.Wad(ensure_wad(), META_PATH)
    for index in range(golden.shape[0]):
        pos, yaw = wad.build_level(index).start()
        mv, pr = camera.pose_from_player(pos, yaw, 1e-8, 320, 200)
        assert np.array_equal(golden[index, 0, :16], mv) and np.array_equal(golden[index, 0, 16:32], pr), index


def test_reset_poses_against_the_float64_composition():
    worst = 0.0
    for k in range(16):
        yaw = np.float32(k * np.pi / 8 + 0.013 * k)
        for pos in ((1.5, 0.2, -3.0), (-27.84, 0.5, -17.28), (10.24, 0.0, 5.12)):
            a = rd.pose_from_player(pos, yaw, 1e-8, 1920, 1080)
            want = _from_player_f64(np.asarray(pos, np.float32), float(yaw), float(np.float32(1e-8)))
            worst = max(worst, float(np.abs(a['modelview'] - want).max()))
            b = rd.pose_look((pos[0], np.float32(pos[1]) + np.float32(0.12), pos[2]), yaw, 1e-8, 1920, 1080)
            assert np.abs(a['modelview'] - b['modelview']).max() <= 1.6e-5
    assert worst <= 1.6e-5, worst


def test_euler_pitch_follows_the_conversion_the_reference_calls():
    """away from a reset the reference composes rotations incrementally (player.rs:215-217); the helper keeps what
    Player::reset does -- Quaternion::from(Euler) -- for any pitch: checked against the float64 form of the same formula"""
    for yaw, pitch in ((0.7, 0.3), (2.5, -0.4), (5.9, 0.1)):
        a = rd.pose_from_player((1.5, 0.2, -3.0), yaw, pitch, 640, 400)
        want = _from_player_f64(np.asarray((1.5, 0.2, -3.0), np.float32), float(np.float32(yaw)), float(np.float32(pitch)))
        assert np.abs(a['modelview'] - want).max() <= 2e-6
        m = a['modelview'].reshape(4, 4).T[:3, :3].astype(np.float64)
        assert np.abs(m @ m.T - np.eye(3)).max() <= 1e-6   # a rotation
