// Host-side loader: the C++ counterpart of rust-doom's `wad` crate (the API surface this project
// keeps).  Names and behaviour follow the reference; citations are file:line into /root/reference.
//
//   WadName            wad/src/name.rs          Archive / LumpReader   wad/src/archive.rs
//   Level              wad/src/level.rs         Image                  wad/src/image.rs
//   TextureDirectory   wad/src/tex.rs           WadMetadata            wad/src/meta.rs
//   LightInfo          wad/src/light.rs         LevelVisitor/Walker    wad/src/visitor.rs
//
// All float arithmetic is binary32 in the reference's evaluation order; this library is compiled
// with -ffp-contract=off (Rust never contracts a*b+c).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <optional>
#include <regex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../common.hpp"

namespace rdoom::wad {

struct WadError : std::runtime_error {
  rdoom_status code;
  WadError(rdoom_status c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// ---- name.rs -------------------------------------------------------------------------------------
struct WadName {
  std::array<uint8_t, 8> b{};
  static bool valid_byte(uint8_t up);
  // WadName::from_bytes (name.rs:41-75); throws WadError(RDOOM_CORRUPT_WAD)
  static WadName from_bytes(const uint8_t *value, size_t len);
  static WadName from_str(const std::string &s) { return from_bytes((const uint8_t *)s.data(), s.size()); }
  bool push(uint8_t byte);  // name.rs:17-39; false on error
  bool operator==(const WadName &o) const { return b == o.b; }
  bool operator!=(const WadName &o) const { return b != o.b; }
  bool operator<(const WadName &o) const { return b < o.b; }
  std::string str() const;
  bool is_untextured() const { return b[0] == '-' && b[1] == 0; }                  // util.rs:4-6
  bool is_sky_flat() const { return std::string((const char *)b.data(), 8) == std::string("F_SKY1\0\0", 8); }  // util.rs:8-10
};
struct WadNameHash {
  size_t operator()(const WadName &n) const {
    uint64_t v;
    std::memcpy(&v, n.b.data(), 8);
    return std::hash<uint64_t>()(v);
  }
};

// Insertion-ordered map with "insert keeps the first position, replaces the value" semantics of
// indexmap::IndexMap (used by the reference wherever iteration order reaches the output).
template <class V>
class NameIndexMap {
 public:
  void insert(const WadName &k, V v) {
    auto it = index_.find(k);
    if (it == index_.end()) {
      index_.emplace(k, items_.size());
      items_.emplace_back(k, std::move(v));
    } else {
      items_[it->second].second = std::move(v);
    }
  }
  const V *get(const WadName &k) const {
    auto it = index_.find(k);
    return it == index_.end() ? nullptr : &items_[it->second].second;
  }
  const std::vector<std::pair<WadName, V>> &items() const { return items_; }
  size_t size() const { return items_.size(); }

 private:
  std::vector<std::pair<WadName, V>> items_;
  std::unordered_map<WadName, size_t, WadNameHash> index_;
};

// ---- util.rs -------------------------------------------------------------------------------------
struct Pnt2f {
  float x, y;
};
inline float from_wad_height(int16_t x) { return (float)x / 100.0f; }
inline float to_wad_height(float x) { return x * 100.0f; }
inline Pnt2f from_wad_coords(int16_t x, int16_t y) { return {-from_wad_height(y), -from_wad_height(x)}; }

// ---- types.rs (on-disk records, little endian, packed) --------------------------------------------
struct WadThing {
  int16_t x, y, angle;
  uint16_t thing_type, flags;
};
struct WadVertex {
  int16_t x, y;
};
struct WadLinedef {
  uint16_t start_vertex, end_vertex, flags, special_type, sector_tag;
  int16_t right_side, left_side;
  bool impassable() const { return flags & 0x0001; }
  bool upper_unpegged() const { return flags & 0x0008; }
  bool lower_unpegged() const { return flags & 0x0010; }
};
struct WadSidedef {
  int16_t x_offset, y_offset;
  WadName upper_texture, lower_texture, middle_texture;
  uint16_t sector;
};
struct WadSector {
  int16_t floor_height, ceiling_height;
  WadName floor_texture, ceiling_texture;
  int16_t light;
  uint16_t sector_type, tag;
};
struct WadSubsector {
  uint16_t num_segs, first_seg;
};
struct WadSeg {
  uint16_t start_vertex, end_vertex, angle, linedef, direction, offset;
};
struct WadNode {
  int16_t line_x, line_y, step_x, step_y;
  int16_t bbox[8];
  uint16_t right, left;
};

// ---- meta.rs -------------------------------------------------------------------------------------
enum class HeightRef { LowestFloor, NextFloor, HighestFloor, LowestCeiling, HighestCeiling, Floor, Ceiling };
struct HeightDef {
  HeightRef to;
  int16_t offset = 0;
};
struct HeightEffectDef {
  HeightDef first;
  std::optional<HeightDef> second;
};
struct MoveEffectDef {
  std::optional<HeightEffectDef> floor, ceiling;
  bool repeat = false;
  float wait = 0.0f, speed = 0.0f;
};
struct LinedefMetadata {
  uint16_t special_type = 0;
  std::string trigger;
  bool monsters = false, only_once = false;
  std::optional<MoveEffectDef> move_effect;
  std::optional<std::string> exit_effect;
};
struct SkyMetadata {
  WadName texture_name;
  std::string pattern_text;
  std::regex level_pattern;
  float tiled_band_size;
};
struct ThingMetadata {
  uint16_t thing_type;
  WadName sprite;
  std::string sequence;
  bool hanging;
  uint32_t radius;
};
struct WadMetadata {
  std::vector<SkyMetadata> sky;
  std::vector<std::vector<WadName>> animated_flats, animated_walls;
  std::vector<ThingMetadata> things;  // decorations, weapons, powerups, artifacts, ammo, keys, monsters (meta.rs:173-206)
  std::map<uint16_t, LinedefMetadata> linedef;
  static WadMetadata from_file(const std::string &path);  // meta.rs:143-154
  static WadMetadata from_text(const std::string &text);
  const ThingMetadata *find_thing(uint16_t thing_type) const;
  const SkyMetadata *sky_for(const WadName &level_name) const;  // meta.rs:156-171
};

// ---- archive.rs ----------------------------------------------------------------------------------
struct LumpInfo {
  WadName name;
  uint64_t offset;
  size_t size;
  bool outside = false;  // the directory entry points outside the file: an error when (and only when) the lump is read
};
class Archive {
 public:
  static std::unique_ptr<Archive> open(const std::string &wad_path, const std::string &meta_path);  // archive.rs:36-60
  const WadMetadata &metadata() const { return meta_; }
  size_t num_levels() const { return levels_.size(); }
  size_t num_lumps() const { return lumps_.size(); }
  size_t level_lump_index(size_t level) const;
  const LumpInfo &lump(size_t index) const;                 // lump_by_index; throws on missing
  std::optional<size_t> named_lump(const WadName &n) const;  // archive.rs:129-139
  size_t required_named_lump(const char *name) const;        // archive.rs:118-127
  // LumpReader::read (archive.rs:244-257): the reference seeks and reads a lump when it is asked for, so a directory entry
  // that points outside the file fails THEN (ErrorKind::Io, errors.rs:87-93) -- never for a lump nobody reads
  const uint8_t *lump_data(size_t index) const;
  // LumpReader::decode_vec (archive.rs:172-190): size must be a positive multiple of `record`
  size_t checked_count(size_t index, size_t record) const;

 private:
  std::vector<uint8_t> data_;
  std::vector<LumpInfo> lumps_;
  std::unordered_map<WadName, size_t, WadNameHash> index_map_;
  std::vector<size_t> levels_;
  WadMetadata meta_;
};

// ---- level.rs ------------------------------------------------------------------------------------
struct NeighbourHeights {
  int16_t lowest_floor, highest_floor, lowest_ceiling, highest_ceiling;
  std::optional<int16_t> next_floor;
};
class Level {
 public:
  static Level from_archive(const Archive &wad, size_t index);  // level.rs:34-81
  std::vector<WadThing> things;
  std::vector<WadLinedef> linedefs;
  std::vector<WadSidedef> sidedefs;
  std::vector<WadVertex> vertices;
  std::vector<WadSeg> segs;
  std::vector<WadSubsector> subsectors;
  std::vector<WadNode> nodes;
  std::vector<WadSector> sectors;
  WadName name;

  std::optional<Pnt2f> vertex(uint16_t id) const;
  const WadLinedef *seg_linedef(const WadSeg &s) const;
  const WadSidedef *side(int16_t index) const;  // left_sidedef/right_sidedef (level.rs:139-151)
  const WadSidedef *seg_sidedef(const WadSeg &s) const;
  const WadSidedef *seg_back_sidedef(const WadSeg &s) const;
  const WadSector *sidedef_sector(const WadSidedef *s) const;
  const WadSector *seg_sector(const WadSeg &s) const { return sidedef_sector(seg_sidedef(s)); }
  const WadSector *seg_back_sector(const WadSeg &s) const { return sidedef_sector(seg_back_sidedef(s)); }
  uint16_t sector_id(const WadSector *s) const { return (uint16_t)(s - sectors.data()); }
  template <class F>
  void for_adjacent_sectors(const WadSector *of, F f) const;  // level.rs:230-258
  int16_t sector_min_light(const WadSector *of) const;         // level.rs:178-182
  std::optional<NeighbourHeights> neighbour_heights(const WadSector *of) const;  // level.rs:184-212
};

// ---- image.rs ------------------------------------------------------------------------------------
constexpr size_t MAX_IMAGE_SIZE = 4096;
class Image {
 public:
  Image() = default;
  Image(size_t w, size_t h, uint16_t fill = 0xFF00);          // Image::new (image.rs:19-32)
  static Image from_buffer(const uint8_t *buf, size_t len);   // image.rs:39-169
  void blit(const Image &src, long ox, long oy, bool ignore_transparency);  // image.rs:171-252
  size_t width() const { return w_; }
  size_t height() const { return h_; }
  const std::vector<uint16_t> &pixels() const { return px_; }
  std::vector<uint16_t> &pixels() { return px_; }
  long x_offset = 0, y_offset = 0;

 private:
  size_t w_ = 0, h_ = 0;
  std::vector<uint16_t> px_;
};

// ---- tex.rs --------------------------------------------------------------------------------------
struct Bounds {
  float pos[2];
  float size[2];
  size_t num_frames;
  size_t row_height;
};
using BoundsLookup = NameIndexMap<Bounds>;
struct TransparentImage {
  std::vector<uint16_t> pixels;
  size_t w = 0, h = 0;
};
struct OpaqueImage {
  std::vector<uint8_t> pixels;
  size_t w = 0, h = 0;
};
class TextureDirectory {
 public:
  static TextureDirectory from_archive(const Archive &wad);  // tex.rs:53-107
  const Image *texture(const WadName &n) const { return textures_.get(n); }
  const std::vector<uint8_t> *flat(const WadName &n) const { return flats_.get(n); }
  size_t num_palettes() const { return palettes_.size() / 768; }
  size_t num_colormaps() const { return colormaps_.size() / 256; }
  const uint8_t *palette(size_t i) const { return palettes_.data() + 768 * i; }
  const uint8_t *colormap(size_t i) const { return colormaps_.data() + 256 * i; }
  std::vector<uint8_t> build_palette_texture(size_t palette, size_t cm_start, size_t cm_end) const;  // tex.rs:137-166
  std::pair<TransparentImage, BoundsLookup> build_texture_atlas(const std::vector<WadName> &names) const;  // tex.rs:168-271
  std::pair<OpaqueImage, BoundsLookup> build_flat_atlas(const std::vector<WadName> &names) const;          // tex.rs:273-333
  size_t num_patches() const { return patches_.size(); }
  size_t num_textures() const { return textures_.size(); }
  size_t num_flats() const { return flats_.size(); }

 private:
  NameIndexMap<Image> textures_;
  std::vector<std::pair<WadName, std::optional<Image>>> patches_;
  std::vector<uint8_t> palettes_, colormaps_;
  NameIndexMap<std::vector<uint8_t>> flats_;
  std::vector<std::vector<WadName>> animated_walls_, animated_flats_;
  void read_patches(const Archive &wad);
  void read_textures(const uint8_t *buf, size_t len);
};

// ---- light.rs ------------------------------------------------------------------------------------
enum class LightEffectKind { Glow, Random, Alternate };
struct LightEffect {
  float alt_level, speed, duration, sync;
  LightEffectKind kind;
  bool operator==(const LightEffect &o) const {
    return alt_level == o.alt_level && speed == o.speed && duration == o.duration && sync == o.sync && kind == o.kind;
  }
};
struct LightInfo {
  float level;
  std::optional<LightEffect> effect;
  bool operator==(const LightInfo &o) const { return level == o.level && effect == o.effect; }
};
LightInfo new_light(const Level &level, const WadSector *sector);  // light.rs:27-79
enum class Contrast { Darken, Brighten };
LightInfo with_contrast(const LightInfo &info, Contrast c);  // light.rs:81-91

// ---- math/src/line.rs ----------------------------------------------------------------------------
struct Line2f {
  Pnt2f origin;
  Pnt2f displace;
  float length;
  static Line2f from_two_points(Pnt2f origin, Pnt2f towards);  // line.rs:12-33
  Line2f inverted_halfspaces() const { return {origin, {-displace.x, -displace.y}, length}; }
  float signed_distance(Pnt2f to) const {  // line.rs:43-45
    return (to.x * displace.y - to.y * displace.x) + (displace.x * origin.y - displace.y * origin.x);
  }
  std::optional<Pnt2f> intersect_point(const Line2f &other) const;  // line.rs:68-84
};

// ---- visitor.rs ----------------------------------------------------------------------------------
struct ObjectId {
  uint32_t v = 0;
};
struct StaticQuad {  // visitor.rs:24-34
  ObjectId object_id;
  Pnt2f v1, v2;
  float tex_start[2], tex_end[2], height_range[2];
  const LightInfo *light_info;
  float scroll;
  std::optional<WadName> tex_name;
  bool blocker;
};
struct StaticPoly {  // visitor.rs:36-42
  ObjectId object_id;
  const Pnt2f *vertices;
  size_t n_vertices;
  float height;
  const LightInfo *light_info;
  WadName tex_name;
};
struct SkyQuad {  // visitor.rs:44-48
  ObjectId object_id;
  Pnt2f v1, v2;
  float height_range[2];
};
struct SkyPoly {  // visitor.rs:50-54
  ObjectId object_id;
  const Pnt2f *vertices;
  size_t n_vertices;
  float height;
};
struct Decor {  // visitor.rs:56-63
  ObjectId object_id;
  float low[3], high[3];
  float half_width;
  const LightInfo *light_info;
  WadName tex_name;
};
enum class Branch { Positive, Negative };
enum class MarkerKind { StartPos, TeleportStart, TeleportEnd };
struct Marker {
  MarkerKind kind;
  size_t player = 0;
};

// `trait LevelVisitor` (visitor.rs:65-127): the reference's own extension point.  Payloads are
// borrowed for the duration of the call -- copy out what you keep.
class LevelVisitor {
 public:
  virtual ~LevelVisitor() = default;
  virtual void visit_wall_quad(const StaticQuad &) {}
  virtual void visit_floor_poly(const StaticPoly &) {}
  virtual void visit_ceil_poly(const StaticPoly &) {}
  virtual void visit_floor_sky_poly(const SkyPoly &) {}
  virtual void visit_ceil_sky_poly(const SkyPoly &) {}
  virtual void visit_sky_quad(const SkyQuad &) {}
  virtual void visit_marker(const float /*pos*/[3], float /*yaw_rad*/, Marker) {}
  virtual void visit_decor(const Decor &) {}
  virtual void visit_bsp_root(const Line2f &) {}
  virtual void visit_bsp_node(const Line2f &, Branch) {}
  virtual void visit_bsp_leaf(Branch) {}
  virtual void visit_bsp_leaf_end() {}
  virtual void visit_bsp_node_end() {}
};

// LevelVisitor::chain (visitor.rs:118-127, 1261-1331)
class VisitorChain : public LevelVisitor {
 public:
  VisitorChain(LevelVisitor &first, LevelVisitor &second) : a_(first), b_(second) {}
  void visit_wall_quad(const StaticQuad &q) override { a_.visit_wall_quad(q), b_.visit_wall_quad(q); }
  void visit_floor_poly(const StaticPoly &p) override { a_.visit_floor_poly(p), b_.visit_floor_poly(p); }
  void visit_ceil_poly(const StaticPoly &p) override { a_.visit_ceil_poly(p), b_.visit_ceil_poly(p); }
  void visit_floor_sky_poly(const SkyPoly &p) override { a_.visit_floor_sky_poly(p), b_.visit_floor_sky_poly(p); }
  void visit_ceil_sky_poly(const SkyPoly &p) override { a_.visit_ceil_sky_poly(p), b_.visit_ceil_sky_poly(p); }
  void visit_sky_quad(const SkyQuad &q) override { a_.visit_sky_quad(q), b_.visit_sky_quad(q); }
  void visit_marker(const float pos[3], float yaw, Marker m) override { a_.visit_marker(pos, yaw, m), b_.visit_marker(pos, yaw, m); }
  void visit_decor(const Decor &d) override { a_.visit_decor(d), b_.visit_decor(d); }
  void visit_bsp_root(const Line2f &l) override { a_.visit_bsp_root(l), b_.visit_bsp_root(l); }
  void visit_bsp_node(const Line2f &l, Branch br) override { a_.visit_bsp_node(l, br), b_.visit_bsp_node(l, br); }
  void visit_bsp_leaf(Branch br) override { a_.visit_bsp_leaf(br), b_.visit_bsp_leaf(br); }
  void visit_bsp_leaf_end() override { a_.visit_bsp_leaf_end(), b_.visit_bsp_leaf_end(); }
  void visit_bsp_node_end() override { a_.visit_bsp_node_end(), b_.visit_bsp_node_end(); }

 private:
  LevelVisitor &a_, &b_;
};

struct DynamicSectorInfo {  // visitor.rs:158-165
  ObjectId floor_id, ceiling_id;
  std::optional<NeighbourHeights> neighbour_heights;
  std::optional<std::pair<int16_t, int16_t>> floor_range, ceiling_range;
};

class LevelAnalysis {  // visitor.rs:316-497
 public:
  LevelAnalysis(const Level &level, const WadMetadata &meta);
  size_t num_objects() const { return num_objects_; }
  size_t num_triggers() const { return num_triggers_; }
  const DynamicSectorInfo *dynamic(uint16_t sector_id) const {
    auto it = dynamic_info_.find(sector_id);
    return it == dynamic_info_.end() ? nullptr : &it->second;
  }

 private:
  std::map<uint16_t, DynamicSectorInfo> dynamic_info_;
  size_t num_objects_ = 0, num_triggers_ = 0;
};

struct SectorInfo {  // visitor.rs:145-156
  ObjectId floor_id, ceiling_id;
  std::pair<int16_t, int16_t> floor_range, ceiling_range;
  int16_t max_height() const { return (int16_t)(ceiling_range.second - floor_range.first); }
};

// ---- SEG -> wall quads on the device (csrc/hip/tessellate.hip) --------------------------------------
// SegInput: everything LevelWalker::seg + wall_quad + sky_quad (visitor.rs:711-937, 987-1008) read for one seg,
// with every table lookup resolved on the host.  SegGeometry: the quads they emit, in emission order.
// Plain data shared by the host walk and the HIP kernel.
struct SegInput {
  float v1x, v1y, v2x, v2y;                      // seg end points, world units (from_wad_coords)
  int16_t floor, ceiling, back_floor, back_ceiling;
  int16_t f_floor_lo, f_floor_hi, f_ceil_lo, f_ceil_hi;  // SectorInfo ranges of the sub-sector's sector
  int16_t b_floor_lo, b_floor_hi, b_ceil_lo, b_ceil_hi;  // ... of the back sector
  int16_t mn, mx;                                // level height range (visitor.rs:1173-1182)
  int16_t x_offset, y_offset;                    // sidedef offsets
  uint16_t seg_offset;
  int16_t tex_h[3];                              // texture height of lower, upper, middle; -1 untextured, -2 unknown texture
  uint16_t sector_height_pad;
  uint32_t flags;                                // SEG_* bits
  uint32_t f_floor_id, f_ceil_id, b_floor_id, b_ceil_id;
};
constexpr uint32_t SEG_VALID = 1u, SEG_HAS_BACK = 2u, SEG_UNPEG_LOWER = 4u, SEG_UNPEG_UPPER = 8u, SEG_IMPASSABLE = 16u,
                   SEG_SCROLL = 32u, SEG_F_CEIL_SKY = 64u, SEG_F_FLOOR_SKY = 128u, SEG_B_CEIL_SKY = 256u,
                   SEG_B_FLOOR_SKY = 512u, SEG_LIGHT_EFFECT = 1024u;
struct SegQuadGeometry {  // one StaticQuad minus the host-side references (light, texture name)
  uint32_t valid, object_id;
  float v1x, v1y, v2x, v2y, s1, t1, s2, t2, low, high, scroll;
  uint32_t contrast;  // 0 none, 1 brighten, 2 darken (visitor.rs:889-901)
  uint32_t slot;      // 0 lower, 1 upper, 2 middle texture of the sidedef
  uint32_t blocker;
};
struct SegSkyGeometry {
  uint32_t valid, object_id;
  float v1x, v1y, v2x, v2y, low, high;
};
struct SegGeometry {
  SegQuadGeometry quad[3];  // one-sided: [0] = the wall; two-sided: lower, upper, middle
  SegSkyGeometry sky[2];    // ceiling side, floor side
};

// points_to_polygon (visitor.rs:1192-1259); exposed because the GPU tessellation kernel restates it.
void points_to_polygon(std::vector<Pnt2f> &points);
Line2f partition_line(const WadNode &node);  // visitor.rs:1150-1155

class LevelWalker {  // visitor.rs:499-1138
 public:
  LevelWalker(const Level &level, const LevelAnalysis &analysis, const TextureDirectory &tex, const WadMetadata &meta,
              LevelVisitor &visitor);
  void walk();  // visitor.rs:541-555

  // Optional hook used by the GPU tessellation path: when set, sub-sector polygons are taken from
  // this table (indexed by sub-sector id) instead of being computed on the host.
  const std::vector<std::vector<Pnt2f>> *precomputed_polygons = nullptr;
  // Records, per visited sub-sector, the BSP half-plane stack (inputs of the tessellation kernel).
  struct LeafInput {
    uint32_t subsector;
    std::vector<Line2f> bsp_lines;
  };
  std::vector<LeafInput> *record_leaves = nullptr;
  // Same for the SEG -> wall-quad half: record_segs (sized level.segs.size()) collects the per-seg inputs
  // instead of emitting quads; precomputed_segs (same indexing) replaces the host arithmetic of seg() /
  // wall_quad() / sky_quad() by the device's results.
  std::vector<SegInput> *record_segs = nullptr;
  const std::vector<SegGeometry> *precomputed_segs = nullptr;

 private:
  const Level &level_;
  const LevelAnalysis &analysis_;
  const TextureDirectory &tex_;
  const WadMetadata &meta_;
  LevelVisitor &visitor_;
  std::pair<int16_t, int16_t> height_range_;
  std::vector<Line2f> bsp_lines_;
  std::vector<Pnt2f> subsector_points_;
  std::vector<Line2f> subsector_seg_lines_;
  std::map<uint16_t, LightInfo> light_cache_;

  SectorInfo sector_info(const WadSector *s) const;
  const LightInfo *light_info(const WadSector *s);
  void node(uint16_t id, Branch branch);
  void children(const WadNode &node, const Line2f &partition);
  void subsector(size_t id);
  enum class Peg { Top, Bottom, BottomLower, TopFloat, BottomFloat };
  struct InternalWallQuad {
    ObjectId object_id;
    const WadSector *sector;
    const WadSeg *seg;
    Pnt2f v1, v2;
    int16_t low, high;
    WadName texture_name;
    Peg peg;
    bool blocker;
  };
  void seg(const WadSector *sector, const SectorInfo &info, const WadSeg &seg, Pnt2f v1, Pnt2f v2);
  bool seg_input(const WadSector *sector, const SectorInfo &info, const WadSeg &seg, Pnt2f v1, Pnt2f v2, SegInput &out);
  void emit_seg_geometry(const WadSector *sector, const WadSeg &seg, const SegGeometry &g);
  void wall_quad(const InternalWallQuad &q);
  void flat_poly(const WadSector *sector, const SectorInfo &info);
  void sky_quad(ObjectId id, Pnt2f v1, Pnt2f v2, int16_t low, int16_t high);
  void things();
  const WadSector *sector_at(Pnt2f pos) const;
  void decor(const WadThing &thing, Pnt2f pos, const WadSector *sector);
};

// helpers shared with the tessellation kernel's host side
inline float magnitude(float x, float y) { return std::sqrt(x * x + y * y); }
inline Pnt2f normalize_or_zero(float x, float y) {  // math/src/lib.rs:36-47
  const float eps = 1.1920929e-7f;
  float m = magnitude(x, y);
  if (!(m > eps)) m = eps;  // f32::max(m, EPSILON)
  return {x / m, y / m};
}
constexpr float BSP_TOLERANCE = 1e-3f;
constexpr float SEG_TOLERANCE = 0.1f;
constexpr float POLY_BIAS = 0.64f * 3e-4f;

}  // namespace rdoom::wad
