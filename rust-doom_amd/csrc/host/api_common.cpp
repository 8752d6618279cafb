// C ABI pieces that do not belong to a subsystem: last-error storage and the camera helper.
#include <cmath>
#include <cstring>
#include <mutex>

#include "../common.hpp"

namespace rdoom {
std::string &last_error_ref() {
  static thread_local std::string err;
  return err;
}
namespace {
std::mutex g_debug_mutex;
DebugOptions g_debug_options;
}  // namespace
DebugOptions debug_options() {
  std::lock_guard<std::mutex> lock(g_debug_mutex);
  return g_debug_options;
}
}  // namespace rdoom

extern "C" {

rdoom_status rdoom_debug_set(const char *name, int32_t value) {
  if (!name) return rdoom::fail(RDOOM_BAD_ARG, "name is null");
  std::lock_guard<std::mutex> lock(rdoom::g_debug_mutex);
  rdoom::DebugOptions &o = rdoom::g_debug_options;
  const struct {
    const char *name;
    int *field;
  } table[] = {{"no_bins", &o.no_bins},     {"entry_cap", &o.entry_cap},   {"vis32", &o.vis32},
               {"leak_mod", &o.leak_mod},   {"frag_nq", &o.frag_nq},       {"frag_bw", &o.frag_bw},
               {"frag_chunk", &o.frag_chunk}, {"bin_threads", &o.bin_threads}, {"no_cover", &o.no_cover},
               {"raster_stats", &o.raster_stats}, {"no_qtab", &o.no_qtab},
               {"keep_vis", &o.keep_vis},   {"qpath", &o.qpath},           {"no_split", &o.no_split},
               {"no_pair", &o.no_pair},     {"no_settle", &o.no_settle},   {"settle_max", &o.settle_max}};
  for (const auto &t : table)
    if (std::strcmp(t.name, name) == 0) {
      *t.field = value;
      return RDOOM_OK;
    }
  if (std::strcmp(name, "reset") == 0) {
    o = rdoom::DebugOptions{};
    return RDOOM_OK;
  }
  return rdoom::fail(RDOOM_BAD_ARG, "unknown debug option '%s'", name);
}

const char *rdoom_last_error(void) { return rdoom::last_error_ref().c_str(); }

// Camera of the reference (game/src/player.rs:84-89, 325-345; engine/src/projections.rs:93-101;
// engine/src/renderer.rs:78-87).  The matrices are *inputs* of the renderer, so this helper may use
// double precision internally; what matters is that every consumer is handed the same 32 floats.
rdoom_status rdoom_pose_look(const float eye[3], float yaw, float pitch, uint32_t width, uint32_t height, float time,
                             rdoom_pose *out) {
  if (!eye || !out || width == 0 || height == 0) return rdoom::fail(RDOOM_BAD_ARG, "bad argument");
  const double cy = std::cos((double)yaw), sy = std::sin((double)yaw), cp = std::cos((double)pitch),
               sp = std::sin((double)pitch);
  // R = Ry(yaw) * Rx(pitch); view = [R^T | -R^T eye]
  const double R[3][3] = {{cy, sy * sp, sy * cp}, {0.0, cp, -sp}, {-sy, cy * sp, cy * cp}};
  std::memset(out, 0, sizeof *out);
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) out->modelview[c * 4 + r] = (float)R[c][r];  // (R^T)[r][c] = R[c][r]
  for (int r = 0; r < 3; r++) {
    double t = 0;
    for (int k = 0; k < 3; k++) t -= R[k][r] * (double)eye[k];
    out->modelview[12 + r] = (float)t;
  }
  out->modelview[15] = 1.0f;
  const float fovy = 65.0f, near_ = 0.01f, far_ = 100.0f;
  const float aspect = ((float)width / (float)height) * 1.2f;
  const float f = (float)(1.0 / std::tan((double)fovy * M_PI / 360.0));
  out->projection[0] = f / aspect;
  out->projection[5] = f;
  out->projection[10] = (far_ + near_) / (near_ - far_);
  out->projection[11] = -1.0f;
  out->projection[14] = (2.0f * far_ * near_) / (near_ - far_);
  out->time = time;
  return RDOOM_OK;
}

// The reference's own arithmetic for the same camera, in binary32 throughout (cgmath 0.18.0, Cargo.lock:178 -- a third-party
// crate that is not under /root/reference: its published source is restated here, call sites cited):
//   player transform  Decomposed { scale 1, rot = Quaternion::from(Euler { x: pitch, y: yaw, z: 0 }), disp = pos }
//                     (game/src/player.rs:124-131: pitch = Rad(1e-8) at a reset, yaw = level.start_yaw(), pos = level.start_pos())
//   camera transform  Decomposed { scale 1, rot = identity, disp = (0, camera_height, 0) } as a child (player.rs:325-335)
//   absolute          player.concat(camera) (engine/src/transforms.rs:121)
//   view              absolute.inverse_transform(), Matrix4::from(view) (engine/src/renderer.rs:78-87)
//   projection        cgmath::perspective(Rad::from(Deg(65)), aspect * 1.2, 0.01, 100) (player.rs:336-344, projections.rs:93-101)
// cgmath: Quaternion::from(Euler) is the euclideanspace.net conversion on half angles; v * q rotates by
// tmp = q.v x v + v * q.s, result = (q.v x tmp) * 2 + v; Decomposed::concat: rot = a.rot * b.rot, disp = a.rot.rotate(b.disp *
// a.scale) + a.disp; inverse_transform: s = 1 / scale, r = rot.invert() = conjugate / magnitude2, d = r.rotate(disp) * -s;
// Matrix3::from(Quaternion) from the doubled components; Matrix4::from(Decomposed) = (Matrix3 * scale) with w = disp.
namespace {
struct Quat {
  float s, x, y, z;
};
struct V3 {
  float x, y, z;
};
V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
V3 rotate(Quat q, V3 v) {  // impl Mul<Vector3> for Quaternion
  const V3 qv{q.x, q.y, q.z};
  const V3 c = cross(qv, v);
  const V3 tmp{c.x + v.x * q.s, c.y + v.y * q.s, c.z + v.z * q.s};
  const V3 c2 = cross(qv, tmp);
  return {c2.x * 2.0f + v.x, c2.y * 2.0f + v.y, c2.z * 2.0f + v.z};
}
Quat qmul(Quat a, Quat b) {  // impl Mul for Quaternion
  return {a.s * b.s - a.x * b.x - a.y * b.y - a.z * b.z, a.s * b.x + a.x * b.s + a.y * b.z - a.z * b.y,
          a.s * b.y + a.y * b.s + a.z * b.x - a.x * b.z, a.s * b.z + a.z * b.s + a.x * b.y - a.y * b.x};
}
}  // namespace

rdoom_status rdoom_pose_from_player(const float pos[3], float yaw, float pitch, uint32_t width, uint32_t height, float time,
                                    rdoom_pose *out) {
  if (!pos || !out || width == 0 || height == 0) return rdoom::fail(RDOOM_BAD_ARG, "bad argument");
  // Quaternion::from(Euler { x: pitch, y: yaw, z: 0 })
  const float sx = sinf(pitch * 0.5f), cx = cosf(pitch * 0.5f), sy = sinf(yaw * 0.5f), cy = cosf(yaw * 0.5f), sz = sinf(0.0f * 0.5f),
              cz = cosf(0.0f * 0.5f);
  const Quat player{-sx * sy * sz + cx * cy * cz, sx * cy * cz + sy * sz * cx, -sx * sz * cy + sy * cx * cz, sx * sy * cz + sz * cx * cy};
  // absolute = player.concat(camera)
  const Quat identity{1.0f, 0.0f, 0.0f, 0.0f};
  const float scale = 1.0f * 1.0f;
  const Quat rot = qmul(player, identity);
  const V3 cam{0.0f * 1.0f, 0.12f * 1.0f, 0.0f * 1.0f};  // other.disp * self.scale
  const V3 rc = rotate(player, cam);
  const V3 disp{rc.x + pos[0], rc.y + pos[1], rc.z + pos[2]};
  // view = absolute.inverse_transform()
  const float s = 1.0f / scale;
  // Quaternion::magnitude2 = s * s + v.magnitude2(), and Vector3::magnitude2 = dot(v, v) sums its element products FIRST
  // ((x x + y y) + z z) -- not s s + x x + y y + z z left to right, which differs in the last place for some rotations
  const float vv = (rot.x * rot.x + rot.y * rot.y) + rot.z * rot.z;
  const float mag2 = rot.s * rot.s + vv;
  const Quat r{rot.s / mag2, -rot.x / mag2, -rot.y / mag2, -rot.z / mag2};
  const V3 rd = rotate(r, disp);
  const V3 d{rd.x * -s, rd.y * -s, rd.z * -s};
  // Matrix4::from(Decomposed): Matrix3::from(rot) * scale, w = disp
  const float x2 = r.x + r.x, y2 = r.y + r.y, z2 = r.z + r.z;
  const float xx2 = x2 * r.x, xy2 = x2 * r.y, xz2 = x2 * r.z, yy2 = y2 * r.y, yz2 = y2 * r.z, zz2 = z2 * r.z;
  const float sy2 = y2 * r.s, sz2 = z2 * r.s, sx2 = x2 * r.s;
  const float m3[9] = {1.0f - yy2 - zz2, xy2 + sz2, xz2 - sy2, xy2 - sz2, 1.0f - xx2 - zz2, yz2 + sx2, xz2 + sy2, yz2 - sx2, 1.0f - xx2 - yy2};
  std::memset(out, 0, sizeof *out);
  for (int c = 0; c < 3; c++)
    for (int rr = 0; rr < 3; rr++) out->modelview[c * 4 + rr] = m3[c * 3 + rr] * s;
  out->modelview[12] = d.x, out->modelview[13] = d.y, out->modelview[14] = d.z, out->modelview[15] = 1.0f;
  // cgmath::perspective: f = cot(fovy / 2) in binary32, fovy = Rad::from(Deg(65))
  const float fovy = 65.0f * (float)(M_PI / 180.0), near_ = 0.01f, far_ = 100.0f;
  const float aspect = ((float)width / (float)height) * 1.2f;
  const float f = 1.0f / tanf(fovy / 2.0f);
  out->projection[0] = f / aspect;
  out->projection[5] = f;
  out->projection[10] = (far_ + near_) / (near_ - far_);
  out->projection[11] = -1.0f;
  out->projection[14] = (2.0f * far_ * near_) / (near_ - far_);
  out->time = time;
  return RDOOM_OK;
}

}  // extern "C"
