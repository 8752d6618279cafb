// game::{lights, level::Builder} and the level assembly.  See game_level.hpp for the citations.
#include "game_level.hpp"

#include <cmath>
#include <chrono>
#include <cstring>

namespace rdoom::game {
using namespace rdoom::wad;

// ---- Lights (game/src/lights.rs) ---------------------------------------------------------------------
uint8_t Lights::push(const LightInfo &info) {
  for (size_t i = 0; i < lights_.size(); i++)
    if (lights_[i] == info) return (uint8_t)i;
  if (lights_.size() >= 255) throw WadError(RDOOM_BAD_LEVEL, "more than 255 distinct light infos");  // lights.rs:20
  lights_.push_back(info);
  return (uint8_t)(lights_.size() - 1);
}

namespace {
float fract(float x) { return x - std::floor(x); }
float clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
float noise(float sync, float time) {  // lights.rs:62-64
  return fract(1.0f + std::sin((sync + time / 1000.0f) * 12.9898f + sync * 78.233f) * 43758.547f);
}
float light_level_at(const LightInfo &info, float time) {  // lights.rs:33-59
  if (!info.effect) return info.level;
  const LightEffect &e = *info.effect;
  switch (e.kind) {
    case LightEffectKind::Glow: {
      const float scale = info.level - e.alt_level;
      const float phase = time * e.speed / scale;
      return std::fabs(0.5f - fract(phase)) * 2.0f * scale + e.alt_level;
    }
    case LightEffectKind::Random:
      return noise(e.sync, std::floor(time * e.speed)) < e.duration ? e.alt_level : info.level;
    default:
      return fract(time * e.speed + e.sync * 3.5435f) < e.duration ? e.alt_level : info.level;
  }
}
}  // namespace

void Lights::fill_buffer_at(float time, uint8_t out[256]) const {
  std::memset(out, 0, 256);
  for (size_t i = 0; i < lights_.size() && i < 256; i++)
    out[i] = (uint8_t)(clamp01(light_level_at(lights_[i], time)) * 255.0f);
}

// ---- Builder (game/src/level.rs) ---------------------------------------------------------------------
void Builder::any_quad(size_t new_length, std::vector<uint32_t> &out) {  // level.rs:620-634
  const uint32_t v0 = (uint32_t)new_length - 4;
  for (uint32_t k : {0u, 1u, 3u, 1u, 2u, 3u}) out.push_back(v0 + k);
}

void Builder::any_poly(size_t new_length, size_t poly_length, std::vector<uint32_t> &out) {  // level.rs:636-645
  const uint32_t n = (uint32_t)new_length, v0 = n - (uint32_t)poly_length;
  for (uint32_t v1 = v0, v2 = v0 + 1; v1 < n && v2 < n; v1++, v2++) {  // first triangle is (v0, v0, v0+1)
    out.push_back(v0);
    out.push_back(v1);
    out.push_back(v2);
  }
}

void Builder::wall_vertex(Pnt2f xz, float y, float tu, float tv, uint8_t light, float scroll, const Bounds &b) {
  rdoom_static_vertex v{};
  v.a_pos[0] = xz.x, v.a_pos[1] = y, v.a_pos[2] = xz.y;
  v.a_atlas_uv[0] = b.pos[0], v.a_atlas_uv[1] = b.pos[1];
  v.a_tile_uv[0] = tu, v.a_tile_uv[1] = tv;
  v.a_tile_size[0] = b.size[0], v.a_tile_size[1] = b.size[1];
  v.a_scroll_rate = scroll;
  v.a_row_height = (float)b.row_height;
  v.a_num_frames = (uint8_t)b.num_frames;
  v.a_light = light;
  static_vertices.push_back(v);
}

void Builder::flat_vertex(Pnt2f xz, float y, uint8_t light, const Bounds &b) {  // level.rs:536-549
  wall_vertex(xz, y, -xz.x * 100.0f, -xz.y * 100.0f, light, 0.0f, b);
}

void Builder::visit_wall_quad(const StaticQuad &q) {  // level.rs:650-681
  counters.num_wall_quads++;
  if (!q.tex_name) return;
  const Bounds *b = materials_.walls.get(*q.tex_name);
  if (!b) return;  // "No such wall texture"
  const uint8_t light = lights.push(*q.light_info);
  wall_vertex(q.v1, q.height_range[0], q.tex_start[0], q.tex_start[1], light, q.scroll, *b);
  wall_vertex(q.v2, q.height_range[0], q.tex_end[0], q.tex_start[1], light, q.scroll, *b);
  wall_vertex(q.v2, q.height_range[1], q.tex_end[0], q.tex_end[1], light, q.scroll, *b);
  wall_vertex(q.v1, q.height_range[1], q.tex_start[0], q.tex_end[1], light, q.scroll, *b);
  any_quad(static_vertices.size(), object_indices[q.object_id.v].wall);
}

void Builder::flat_poly_common(const StaticPoly &p, bool reverse) {
  const Bounds *b = materials_.flats.get(p.tex_name);
  if (!b) return;  // "No such floor/ceiling texture"
  const uint8_t light = lights.push(*p.light_info);
  for (size_t i = 0; i < p.n_vertices; i++)
    flat_vertex(p.vertices[reverse ? p.n_vertices - 1 - i : i], p.height, light, *b);
  any_poly(static_vertices.size(), p.n_vertices, object_indices[p.object_id.v].flat);
}

void Builder::visit_floor_poly(const StaticPoly &p) {  // level.rs:683-703
  counters.num_floor_polys++;
  float cx = 0, cz = 0;
  for (size_t i = 0; i < p.n_vertices; i++) cx += p.vertices[i].x, cz += p.vertices[i].y;
  floor_centroids.insert(floor_centroids.end(), {cx / (float)p.n_vertices, p.height, cz / (float)p.n_vertices});
  flat_poly_common(p, false);
}

void Builder::visit_ceil_poly(const StaticPoly &p) {  // level.rs:705-725
  counters.num_ceil_polys++;
  flat_poly_common(p, true);
}

void Builder::sky_poly_common(const SkyPoly &p, bool reverse) {
  for (size_t i = 0; i < p.n_vertices; i++) {
    const Pnt2f v = p.vertices[reverse ? p.n_vertices - 1 - i : i];
    sky_vertices.insert(sky_vertices.end(), {v.x, p.height, v.y});
  }
  any_poly(sky_vertices.size() / 3, p.n_vertices, object_indices[p.object_id.v].sky);
}

void Builder::visit_floor_sky_poly(const SkyPoly &p) {  // level.rs:727-733
  counters.num_sky_floor_polys++;
  sky_poly_common(p, false);
}

void Builder::visit_ceil_sky_poly(const SkyPoly &p) {  // level.rs:735-741
  counters.num_sky_ceil_polys++;
  sky_poly_common(p, true);
}

void Builder::visit_sky_quad(const SkyQuad &q) {  // level.rs:743-755
  counters.num_sky_wall_quads++;
  sky_vertices.insert(sky_vertices.end(), {q.v1.x, q.height_range[0], q.v1.y, q.v2.x, q.height_range[0], q.v2.y,
                                           q.v2.x, q.height_range[1], q.v2.y, q.v1.x, q.height_range[1], q.v1.y});
  any_quad(sky_vertices.size() / 3, object_indices[q.object_id.v].sky);
}

void Builder::visit_marker(const float pos[3], float yaw, Marker marker) {  // level.rs:757-762
  if (marker.kind == MarkerKind::StartPos && marker.player == 0) {
    start_pos[0] = pos[0] + 0.0f;
    start_pos[1] = pos[1] + 0.5f;
    start_pos[2] = pos[2] + 32.0f / 100.0f;
    start_yaw = yaw;
  }
}

void Builder::visit_decor(const Decor &d) {  // level.rs:764-793
  counters.num_decors++;
  const uint8_t light = lights.push(*d.light_info);
  const Bounds *b = materials_.decor.get(d.tex_name);
  if (!b) return;  // "No such decor texture"
  auto vertex = [&](const float pos[3], float local_x, float tu, float tv) {
    rdoom_sprite_vertex v{};
    std::memcpy(v.a_pos, pos, 12);
    v.a_local_x = local_x;
    v.a_atlas_uv[0] = b->pos[0], v.a_atlas_uv[1] = b->pos[1];
    v.a_tile_uv[0] = tu, v.a_tile_uv[1] = tv;
    v.a_tile_size[0] = b->size[0], v.a_tile_size[1] = b->size[1];
    v.a_num_frames = 1;
    v.a_light = light;
    decor_vertices.push_back(v);
  };
  vertex(d.low, -d.half_width, 0.0f, b->size[1]);
  vertex(d.low, d.half_width, b->size[0], b->size[1]);
  vertex(d.high, d.half_width, b->size[0], 0.0f);
  vertex(d.high, -d.half_width, 0.0f, 0.0f);
  any_quad(decor_vertices.size(), object_indices[d.object_id.v].decor);
}

// ---- level assembly -------------------------------------------------------------------------------------
void walk_level(const LoadedWad &w, size_t level_index, wad::LevelVisitor &visitor) {
  const Archive &archive = *w.archive;
  const Level level = Level::from_archive(archive, level_index);
  const LevelAnalysis analysis(level, archive.metadata());
  LevelWalker walker(level, analysis, w.textures, archive.metadata(), visitor);
  walker.walk();
}

std::unique_ptr<BuiltLevel> build_level(const LoadedWad &w, size_t level_index, TessellateFn tessellate,
                                        TessellateSegsFn tessellate_segs, wad::LevelVisitor *chained) {
  const Archive &archive = *w.archive;
  const TextureDirectory &tex = w.textures;
  using Clock = std::chrono::steady_clock;
  auto ms_since = [](Clock::time_point t0) { return std::chrono::duration<float, std::milli>(Clock::now() - t0).count(); };
  auto t0 = Clock::now();
  const Level level = Level::from_archive(archive, level_index);
  const float level_lumps_ms = ms_since(t0);
  t0 = Clock::now();
  const LevelAnalysis analysis(level, archive.metadata());
  const float analysis_ms = ms_since(t0);
  auto out = std::make_unique<BuiltLevel>();
  out->timings.level_lumps_ms = level_lumps_ms;
  out->timings.analysis_ms = analysis_ms;
  t0 = Clock::now();

  // which names feed which atlas (game/src/game_shaders.rs:282-356)
  std::vector<WadName> flat_names, wall_names, decor_names;
  for (const WadSector &s : level.sectors)
    for (const WadName &n : {s.floor_texture, s.ceiling_texture})
      if (!n.is_untextured() && !n.is_sky_flat()) flat_names.push_back(n);
  for (const WadSidedef &s : level.sidedefs)
    for (const WadName &n : {s.upper_texture, s.lower_texture, s.middle_texture})
      if (!n.is_untextured()) wall_names.push_back(n);
  for (const WadThing &t : level.things) {
    const ThingMetadata *m = archive.metadata().find_thing(t.thing_type);
    if (!m) continue;
    WadName sprite0 = m->sprite;
    if (!m->sequence.empty()) (void)sprite0.push((uint8_t)m->sequence[0]);
    WadName sprite1 = sprite0;
    if (sprite0.push('0')) decor_names.push_back(sprite0);
    if (sprite1.push('1')) decor_names.push_back(sprite1);
  }
  LevelMaterials materials;
  {
    auto fa = tex.build_flat_atlas(flat_names);
    out->flat_atlas = std::move(fa.first);
    materials.flats = std::move(fa.second);
    auto wa = tex.build_texture_atlas(wall_names);
    out->wall_atlas = std::move(wa.first);
    materials.walls = std::move(wa.second);
    auto da = tex.build_texture_atlas(decor_names);
    out->decor_atlas = std::move(da.first);
    materials.decor = std::move(da.second);
  }
  // sky uniforms (game_shaders.rs:358-387)
  if (const SkyMetadata *sky = archive.metadata().sky_for(level.name)) {
    out->sky_band = sky->tiled_band_size;
    if (const Image *img = tex.texture(sky->texture_name)) {
      out->sky_texture.w = img->width();
      out->sky_texture.h = img->height();
      out->sky_texture.pixels = img->pixels();
    }
  }
  if (out->sky_texture.pixels.empty()) {  // dummy 1x1 texture (game_shaders.rs:413-419)
    out->sky_texture.w = out->sky_texture.h = 1;
    out->sky_texture.pixels = {0};
  }
  out->playpal.assign(tex.palette(0), tex.palette(0) + 768);
  if (tex.num_colormaps() < 32) throw WadError(RDOOM_CORRUPT_WAD, "COLORMAP has fewer than 32 maps");
  out->colormap.assign(tex.colormap(0), tex.colormap(0) + 32 * 256);

  out->timings.atlases_ms = ms_since(t0);
  t0 = Clock::now();
  Builder builder(materials);
  LevelVisitor no_second;
  VisitorChain chain(builder, chained ? *chained : no_second);  // builder.chain(..): level.rs:378-382
  LevelWalker walker(level, analysis, tex, archive.metadata(), chained ? static_cast<LevelVisitor &>(chain) : builder);
  std::vector<std::vector<Pnt2f>> polygons;
  std::vector<wad::SegGeometry> seg_geometry;
  if (tessellate) {
    // leaf inputs (and per-seg inputs) come from a visitor-free pre-walk; polygons and wall quads from the device
    std::vector<LevelWalker::LeafInput> leaves;
    std::vector<wad::SegInput> seg_inputs(level.segs.size());  // flags == 0: seg never visited / skipped
    {
      LevelVisitor nothing;
      LevelWalker pre(level, analysis, tex, archive.metadata(), nothing);
      pre.record_leaves = &leaves;
      if (tessellate_segs) pre.record_segs = &seg_inputs;
      std::vector<std::vector<Pnt2f>> empty(level.subsectors.size());
      pre.precomputed_polygons = &empty;
      pre.walk();
    }
    polygons = tessellate(level, leaves);
    walker.precomputed_polygons = &polygons;
    if (tessellate_segs) {
      seg_geometry = tessellate_segs(seg_inputs);
      walker.precomputed_segs = &seg_geometry;
    }
  }
  walker.walk();

  out->static_vertices = std::move(builder.static_vertices);
  out->sky_vertices = std::move(builder.sky_vertices);
  out->decor_vertices = std::move(builder.decor_vertices);
  // draw order: per object id ascending: flats, walls, decor, sky (game/src/level.rs:443-496)
  for (auto &kv : builder.object_indices) {
    const Indices &ind = kv.second;
    auto add = [&](uint32_t kind, const std::vector<uint32_t> &src, std::vector<uint32_t> &dst) {
      if (src.empty()) return;
      out->draws.push_back({kind, kv.first, (uint32_t)dst.size(), (uint32_t)src.size()});
      dst.insert(dst.end(), src.begin(), src.end());
    };
    add(RDOOM_KIND_FLAT, ind.flat, out->static_indices);
    add(RDOOM_KIND_WALL, ind.wall, out->static_indices);
    add(RDOOM_KIND_DECOR, ind.decor, out->decor_indices);
    add(RDOOM_KIND_SKY, ind.sky, out->sky_indices);
  }
  out->lights = builder.lights;
  std::memcpy(out->start_pos, builder.start_pos, sizeof out->start_pos);
  out->start_yaw = builder.start_yaw;
  out->counters = builder.counters;
  out->counters.num_static_tris = (uint32_t)(out->static_indices.size() / 3);
  out->counters.num_sky_tris = (uint32_t)(out->sky_indices.size() / 3);
  out->counters.num_sprite_tris = (uint32_t)(out->decor_indices.size() / 3);
  // the reference crashes on levels without a tagged sector: defined as 1 object
  out->counters.num_objects = (uint32_t)std::max<size_t>(1, analysis.num_objects());
  out->counters.num_lights = (uint32_t)out->lights.size();
  out->floor_centroids = std::move(builder.floor_centroids);
  out->timings.walk_ms = ms_since(t0);
  return out;
}

}  // namespace rdoom::game
