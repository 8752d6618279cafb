// C ABI of the loader + builder (include/rdoom.h "loader + builder"): exceptions stop here.
#include <chrono>
#include <cstring>

#include "game_level.hpp"

namespace rdoom::game {
// defined in csrc/hip/tessellate.hip: SSECTOR -> convex polygon on the device
std::vector<std::vector<wad::Pnt2f>> tessellate_on_device(const wad::Level &level,
                                                          const std::vector<wad::LevelWalker::LeafInput> &leaves);
// SEG -> wall / sky quads on the device
std::vector<wad::SegGeometry> tessellate_segs_on_device(const std::vector<wad::SegInput> &inputs);
}  // namespace rdoom::game

struct rdoom_wad {
  rdoom::game::LoadedWad w;
};
struct rdoom_built {
  std::unique_ptr<rdoom::game::BuiltLevel> b;
};

namespace {
// trait LevelVisitor implemented by a table of C callbacks (include/rdoom.h: rdoom_visitor_vtbl)
class CallbackVisitor : public rdoom::wad::LevelVisitor {
 public:
  CallbackVisitor(const rdoom_visitor_vtbl &v, void *user) : v_(v), user_(user) {}
  void visit_wall_quad(const rdoom::wad::StaticQuad &q) override {
    if (!v_.visit_wall_quad) return;
    rdoom_light_info li;
    rdoom_static_quad c{};
    c.object_id = q.object_id.v;
    xz(c.v1, q.v1), xz(c.v2, q.v2);
    for (int i = 0; i < 2; i++) c.tex_start[i] = q.tex_start[i], c.tex_end[i] = q.tex_end[i], c.height_range[i] = q.height_range[i];
    c.light_info = light(li, q.light_info);
    c.scroll = q.scroll;
    c.has_tex_name = q.tex_name ? 1 : 0;
    if (q.tex_name) std::memcpy(c.tex_name, q.tex_name->b.data(), 8);
    c.blocker = q.blocker ? 1 : 0;
    v_.visit_wall_quad(user_, &c);
  }
  void visit_floor_poly(const rdoom::wad::StaticPoly &p) override { poly(v_.visit_floor_poly, p); }
  void visit_ceil_poly(const rdoom::wad::StaticPoly &p) override { poly(v_.visit_ceil_poly, p); }
  void visit_floor_sky_poly(const rdoom::wad::SkyPoly &p) override { sky_poly(v_.visit_floor_sky_poly, p); }
  void visit_ceil_sky_poly(const rdoom::wad::SkyPoly &p) override { sky_poly(v_.visit_ceil_sky_poly, p); }
  void visit_sky_quad(const rdoom::wad::SkyQuad &q) override {
    if (!v_.visit_sky_quad) return;
    rdoom_sky_quad c{};
    c.object_id = q.object_id.v;
    xz(c.v1, q.v1), xz(c.v2, q.v2);
    c.height_range[0] = q.height_range[0], c.height_range[1] = q.height_range[1];
    v_.visit_sky_quad(user_, &c);
  }
  void visit_marker(const float pos[3], float yaw, rdoom::wad::Marker m) override {
    if (v_.visit_marker) v_.visit_marker(user_, pos, yaw, (int32_t)m.kind, (uint32_t)m.player);
  }
  void visit_decor(const rdoom::wad::Decor &d) override {
    if (!v_.visit_decor) return;
    rdoom_light_info li;
    rdoom_decor c{};
    c.object_id = d.object_id.v;
    std::memcpy(c.low, d.low, 12), std::memcpy(c.high, d.high, 12);
    c.half_width = d.half_width;
    c.light_info = light(li, d.light_info);
    std::memcpy(c.tex_name, d.tex_name.b.data(), 8);
    v_.visit_decor(user_, &c);
  }
  void visit_bsp_root(const rdoom::wad::Line2f &l) override {
    if (!v_.visit_bsp_root) return;
    const rdoom_line2f c = line(l);
    v_.visit_bsp_root(user_, &c);
  }
  void visit_bsp_node(const rdoom::wad::Line2f &l, rdoom::wad::Branch b) override {
    if (!v_.visit_bsp_node) return;
    const rdoom_line2f c = line(l);
    v_.visit_bsp_node(user_, &c, (int32_t)b);
  }
  void visit_bsp_leaf(rdoom::wad::Branch b) override {
    if (v_.visit_bsp_leaf) v_.visit_bsp_leaf(user_, (int32_t)b);
  }
  void visit_bsp_leaf_end() override {
    if (v_.visit_bsp_leaf_end) v_.visit_bsp_leaf_end(user_);
  }
  void visit_bsp_node_end() override {
    if (v_.visit_bsp_node_end) v_.visit_bsp_node_end(user_);
  }

 private:
  static void xz(float out[2], rdoom::wad::Pnt2f p) { out[0] = p.x, out[1] = p.y; }
  static rdoom_line2f line(const rdoom::wad::Line2f &l) {
    return rdoom_line2f{{l.origin.x, l.origin.y}, {l.displace.x, l.displace.y}, l.length};
  }
  static const rdoom_light_info *light(rdoom_light_info &out, const rdoom::wad::LightInfo *in) {
    if (!in) return nullptr;
    out = rdoom_light_info{};
    out.level = in->level;
    if (in->effect) {
      out.has_effect = 1;
      out.effect_kind = (int32_t)in->effect->kind;
      out.alt_level = in->effect->alt_level, out.speed = in->effect->speed, out.duration = in->effect->duration,
      out.sync = in->effect->sync;
    }
    return &out;
  }
  void poly(void (*fn)(void *, const rdoom_static_poly *), const rdoom::wad::StaticPoly &p) {
    if (!fn) return;
    static_assert(sizeof(rdoom::wad::Pnt2f) == 2 * sizeof(float), "Pnt2f is two floats");
    rdoom_light_info li;
    rdoom_static_poly c{};
    c.object_id = p.object_id.v;
    c.vertices = reinterpret_cast<const float *>(p.vertices);
    c.n_vertices = (uint32_t)p.n_vertices;
    c.height = p.height;
    c.light_info = light(li, p.light_info);
    std::memcpy(c.tex_name, p.tex_name.b.data(), 8);
    fn(user_, &c);
  }
  void sky_poly(void (*fn)(void *, const rdoom_sky_poly *), const rdoom::wad::SkyPoly &p) {
    if (!fn) return;
    rdoom_sky_poly c{};
    c.object_id = p.object_id.v;
    c.vertices = reinterpret_cast<const float *>(p.vertices);
    c.n_vertices = (uint32_t)p.n_vertices;
    c.height = p.height;
    fn(user_, &c);
  }
  const rdoom_visitor_vtbl v_;
  void *user_;
};

template <class F>
rdoom_status guarded(F f) {
  try {
    return f();
  } catch (const rdoom::wad::WadError &e) {
    return rdoom::fail(e.code, "%s", e.what());
  } catch (const std::bad_alloc &) {
    return rdoom::fail(RDOOM_OOM, "out of host memory");
  } catch (const std::exception &e) {
    return rdoom::fail(RDOOM_BAD_LEVEL, "%s", e.what());
  }
}
}  // namespace

extern "C" {

rdoom_status rdoom_wad_open(const char *wad_path, const char *metadata_path, rdoom_wad **out_wad) {
  if (!wad_path || !metadata_path || !out_wad) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out_wad = nullptr;
  return guarded([&]() -> rdoom_status {
    using Clock = std::chrono::steady_clock;
    auto w = std::make_unique<rdoom_wad>();
    auto t0 = Clock::now();
    w->w.archive = rdoom::wad::Archive::open(wad_path, metadata_path);
    w->w.timings.open_ms = std::chrono::duration<float, std::milli>(Clock::now() - t0).count();
    t0 = Clock::now();
    w->w.textures = rdoom::wad::TextureDirectory::from_archive(*w->w.archive);
    w->w.timings.textures_ms = std::chrono::duration<float, std::milli>(Clock::now() - t0).count();
    *out_wad = w.release();
    return RDOOM_OK;
  });
}

void rdoom_wad_close(rdoom_wad *wad) { delete wad; }

rdoom_status rdoom_wad_timings(const rdoom_wad *wad, rdoom_host_timings *out) {
  if (!wad || !out) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out = wad->w.timings;
  return RDOOM_OK;
}

rdoom_status rdoom_wad_num_levels(const rdoom_wad *wad, uint32_t *out) {
  if (!wad || !out) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out = (uint32_t)wad->w.archive->num_levels();
  return RDOOM_OK;
}

rdoom_status rdoom_wad_level_name(const rdoom_wad *wad, uint32_t index, char out_name[9]) {
  if (!wad || !out_name) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  return guarded([&]() -> rdoom_status {
    const auto &li = wad->w.archive->lump(wad->w.archive->level_lump_index(index));
    std::memcpy(out_name, li.name.b.data(), 8);
    out_name[8] = 0;
    return RDOOM_OK;
  });
}

rdoom_status rdoom_wad_name_from_bytes(const uint8_t *bytes, uint32_t len, uint8_t out[8]) {
  if ((!bytes && len) || !out) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  return guarded([&]() -> rdoom_status {
    const auto n = rdoom::wad::WadName::from_bytes(bytes, len);
    std::memcpy(out, n.b.data(), 8);
    return RDOOM_OK;
  });
}

rdoom_status rdoom_wad_build_level(const rdoom_wad *wad, uint32_t level_index, int32_t use_gpu_tessellation,
                                   rdoom_built **out_built) {
  if (!wad || !out_built) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out_built = nullptr;
  return guarded([&]() -> rdoom_status {
    auto b = std::make_unique<rdoom_built>();
    b->b = rdoom::game::build_level(wad->w, level_index,
                                    use_gpu_tessellation ? &rdoom::game::tessellate_on_device : nullptr,
                                    use_gpu_tessellation ? &rdoom::game::tessellate_segs_on_device : nullptr);
    *out_built = b.release();
    return RDOOM_OK;
  });
}

rdoom_status rdoom_wad_build_level_chained(const rdoom_wad *wad, uint32_t level_index, int32_t use_gpu_tessellation,
                                           const rdoom_visitor_vtbl *visitor, void *user, rdoom_built **out_built) {
  if (!wad || !out_built || !visitor) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out_built = nullptr;
  return guarded([&]() -> rdoom_status {
    CallbackVisitor second(*visitor, user);
    auto b = std::make_unique<rdoom_built>();
    b->b = rdoom::game::build_level(wad->w, level_index,
                                    use_gpu_tessellation ? &rdoom::game::tessellate_on_device : nullptr,
                                    use_gpu_tessellation ? &rdoom::game::tessellate_segs_on_device : nullptr, &second);
    *out_built = b.release();
    return RDOOM_OK;
  });
}

rdoom_status rdoom_wad_walk(const rdoom_wad *wad, uint32_t level_index, const rdoom_visitor_vtbl *visitor, void *user) {
  if (!wad || !visitor) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  return guarded([&]() -> rdoom_status {
    CallbackVisitor v(*visitor, user);
    rdoom::game::walk_level(wad->w, level_index, v);
    return RDOOM_OK;
  });
}

void rdoom_built_destroy(rdoom_built *built) { delete built; }

rdoom_status rdoom_built_desc(const rdoom_built *built, rdoom_level_desc *d) {
  if (!built || !d) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  const rdoom::game::BuiltLevel &b = *built->b;
  std::memset(d, 0, sizeof *d);
  d->static_verts = b.static_vertices.data();
  d->n_static_verts = (uint32_t)b.static_vertices.size();
  d->static_indices = b.static_indices.data();
  d->n_static_indices = (uint32_t)b.static_indices.size();
  d->sky_verts = b.sky_vertices.data();
  d->n_sky_verts = (uint32_t)(b.sky_vertices.size() / 3);
  d->sky_indices = b.sky_indices.data();
  d->n_sky_indices = (uint32_t)b.sky_indices.size();
  d->decor_verts = b.decor_vertices.data();
  d->n_decor_verts = (uint32_t)b.decor_vertices.size();
  d->decor_indices = b.decor_indices.data();
  d->n_decor_indices = (uint32_t)b.decor_indices.size();
  d->draws = b.draws.data();
  d->n_draws = (uint32_t)b.draws.size();
  d->flat_atlas = b.flat_atlas.pixels.data();
  d->flat_w = (uint32_t)b.flat_atlas.w;
  d->flat_h = (uint32_t)b.flat_atlas.h;
  d->wall_atlas = b.wall_atlas.pixels.data();
  d->wall_w = (uint32_t)b.wall_atlas.w;
  d->wall_h = (uint32_t)b.wall_atlas.h;
  d->decor_atlas = b.decor_atlas.pixels.data();
  d->decor_w = (uint32_t)b.decor_atlas.w;
  d->decor_h = (uint32_t)b.decor_atlas.h;
  d->sky_texture = b.sky_texture.pixels.data();
  d->sky_w = (uint32_t)b.sky_texture.w;
  d->sky_h = (uint32_t)b.sky_texture.h;
  d->sky_tiled_band_size = b.sky_band;
  d->playpal = b.playpal.data();
  d->colormap = b.colormap.data();
  return RDOOM_OK;
}

rdoom_status rdoom_built_counters(const rdoom_built *built, rdoom_counters *out) {
  if (!built || !out) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out = built->b->counters;
  return RDOOM_OK;
}

rdoom_status rdoom_built_timings(const rdoom_built *built, rdoom_host_timings *out) {
  if (!built || !out) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out = built->b->timings;
  return RDOOM_OK;
}

rdoom_status rdoom_built_lights_at(const rdoom_built *built, float time, uint8_t out_lights[256]) {
  if (!built || !out_lights) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  built->b->lights.fill_buffer_at(time, out_lights);
  return RDOOM_OK;
}

rdoom_status rdoom_built_start(const rdoom_built *built, float out_pos[3], float *out_yaw) {
  if (!built || !out_pos || !out_yaw) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  std::memcpy(out_pos, built->b->start_pos, 12);
  *out_yaw = built->b->start_yaw;
  return RDOOM_OK;
}

rdoom_status rdoom_built_floor_centroids(const rdoom_built *built, const float **out_xyz, uint32_t *out_n) {
  if (!built || !out_xyz || !out_n) return rdoom::fail(RDOOM_BAD_ARG, "null argument");
  *out_xyz = built->b->floor_centroids.data();
  *out_n = (uint32_t)(built->b->floor_centroids.size() / 3);
  return RDOOM_OK;
}

}  // extern "C"
