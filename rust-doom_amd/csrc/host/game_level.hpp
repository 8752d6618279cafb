// game::{lights, level::Builder}: the `LevelVisitor` that turns walk events into the vertex / index
// arrays rust-doom uploads to GL.  Reference: game/src/lights.rs, game/src/level.rs:275-327,513-794,
// game/src/game_shaders.rs:175-387 (which names feed which atlas), game/src/vertex.rs.
#pragma once
#include <map>
#include <vector>

#include "wad.hpp"

namespace rdoom::game {

class Lights {  // game/src/lights.rs
 public:
  uint8_t push(const wad::LightInfo &info);            // lights.rs:14-24
  void fill_buffer_at(float time, uint8_t out[256]) const;  // lights.rs:26-30
  size_t size() const { return lights_.size(); }

 private:
  std::vector<wad::LightInfo> lights_;
};

struct Indices {
  std::vector<uint32_t> wall, flat, sky, decor;
};

struct LevelMaterials {  // the atlas bounds the Builder looks names up in (game_shaders.rs LevelMaterials)
  wad::BoundsLookup flats, walls, decor;
};

class Builder : public wad::LevelVisitor {  // game/src/level.rs:307-327, 648-794
 public:
  explicit Builder(const LevelMaterials &materials) : materials_(materials) {}
  void visit_wall_quad(const wad::StaticQuad &quad) override;
  void visit_floor_poly(const wad::StaticPoly &poly) override;
  void visit_ceil_poly(const wad::StaticPoly &poly) override;
  void visit_floor_sky_poly(const wad::SkyPoly &poly) override;
  void visit_ceil_sky_poly(const wad::SkyPoly &poly) override;
  void visit_sky_quad(const wad::SkyQuad &quad) override;
  void visit_marker(const float pos[3], float yaw, wad::Marker marker) override;
  void visit_decor(const wad::Decor &decor) override;

  Lights lights;
  float start_pos[3] = {0, 0, 0};
  float start_yaw = 0.0f;
  std::vector<rdoom_static_vertex> static_vertices;
  std::vector<float> sky_vertices;  // xyz triples
  std::vector<rdoom_sprite_vertex> decor_vertices;
  std::map<uint32_t, Indices> object_indices;  // VecMap<Indices>: ascending object id
  rdoom_counters counters{};
  rdoom_host_timings timings{};  // level_lumps_ms, atlases_ms, analysis_ms, walk_ms of this build
  std::vector<float> floor_centroids;  // xyz per floor polygon (pose generators)

 private:
  const LevelMaterials &materials_;
  void wall_vertex(wad::Pnt2f xz, float y, float tu, float tv, uint8_t light, float scroll, const wad::Bounds &b);
  void flat_vertex(wad::Pnt2f xz, float y, uint8_t light, const wad::Bounds &b);
  void flat_poly_common(const wad::StaticPoly &poly, bool reverse);
  void sky_poly_common(const wad::SkyPoly &poly, bool reverse);
  static void any_quad(size_t new_length, std::vector<uint32_t> &out);
  static void any_poly(size_t new_length, size_t poly_length, std::vector<uint32_t> &out);
};

// Everything one level needs on the device, in reference draw order.
struct BuiltLevel {
  std::vector<rdoom_static_vertex> static_vertices;
  std::vector<uint32_t> static_indices;
  std::vector<float> sky_vertices;
  std::vector<uint32_t> sky_indices;
  std::vector<rdoom_sprite_vertex> decor_vertices;
  std::vector<uint32_t> decor_indices;
  std::vector<rdoom_draw> draws;
  wad::OpaqueImage flat_atlas;
  wad::TransparentImage wall_atlas, decor_atlas, sky_texture;
  float sky_band = 0.0f;
  std::vector<uint8_t> playpal, colormap;
  Lights lights;
  float start_pos[3] = {0, 0, 0};
  float start_yaw = 0.0f;
  rdoom_counters counters{};
  rdoom_host_timings timings{};  // level_lumps_ms, atlases_ms, analysis_ms, walk_ms of this build
  std::vector<float> floor_centroids;
};

struct LoadedWad {
  std::unique_ptr<wad::Archive> archive;
  wad::TextureDirectory textures;
  rdoom_host_timings timings{};  // open_ms, textures_ms
};

// WadSystem::create's level half + GameShaders::load_level + Builder::build.
// `tessellate` (optional): called with the level and the recorded BSP leaf inputs, returns the
// sub-sector polygons computed on the GPU (indexed by sub-sector id).
using TessellateFn = std::vector<std::vector<wad::Pnt2f>> (*)(const wad::Level &,
                                                              const std::vector<wad::LevelWalker::LeafInput> &);
// `tessellate_segs` (optional, with `tessellate`): called with the recorded per-seg inputs, returns the wall / sky
// quad geometry computed on the GPU (indexed by seg).
using TessellateSegsFn = std::vector<wad::SegGeometry> (*)(const std::vector<wad::SegInput> &);
// `chained` (optional): a second visitor that sees every event after the Builder (game/src/level.rs:378-382).
std::unique_ptr<BuiltLevel> build_level(const LoadedWad &w, size_t level_index, TessellateFn tessellate,
                                        TessellateSegsFn tessellate_segs = nullptr, wad::LevelVisitor *chained = nullptr);
// WadSystem::walk (game/src/wad_system.rs:47-56): one level, the caller's visitor only.
void walk_level(const LoadedWad &w, size_t level_index, wad::LevelVisitor &visitor);

}  // namespace rdoom::game
