// C ABI of the loader + builder (include/rdoom.h "loader + builder"): exceptions stop here.
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
    auto w = std::make_unique<rdoom_wad>();
    w->w.archive = rdoom::wad::Archive::open(wad_path, metadata_path);
    w->w.textures = rdoom::wad::TextureDirectory::from_archive(*w->w.archive);
    *out_wad = w.release();
    return RDOOM_OK;
  });
}

void rdoom_wad_close(rdoom_wad *wad) { delete wad; }

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
