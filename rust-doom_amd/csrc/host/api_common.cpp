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
               {"keep_vis", &o.keep_vis},   {"qpath", &o.qpath}};
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

}  // extern "C"
