// wad::{light, visitor}: per-sector light info, dynamic-sector analysis and the BSP walk that turns
// SSECTORs into convex floor/ceiling polygons and SEGs into wall quads.
// Reference: wad/src/light.rs, wad/src/visitor.rs, math/src/line.rs.
#include <algorithm>
#include <cmath>

#include "wad.hpp"

namespace rdoom::wad {

// ---- light.rs ------------------------------------------------------------------------------------
namespace {
constexpr uint16_t FLASH = 1, FAST_STROBE_1 = 2, FAST_STROBE_2 = 4, FAST_STROBE_SYNC = 13, SLOW_STROBE = 3,
                   SLOW_STROBE_SYNC = 12, GLOW = 8, FLICKER = 17;
constexpr float F32_EPSILON = 1.1920929e-7f;
float light_to_f32(int16_t level) { return (float)(int16_t)(level >> 3) / 31.0f; }  // light.rs:113-115
float id_to_sync(uint16_t id) {                                                     // light.rs:109-111
  return (float)(((uint64_t)id * 1664525ull + 1013904223ull) & 0xFFFFull) / 15.0f;
}
float clamp01(float v) { return v > 1.0f ? 1.0f : (v < 0.0f ? 0.0f : v); }
}  // namespace

LightInfo new_light(const Level &level, const WadSector *sector) {
  const float base = light_to_f32(sector->light);
  const uint16_t st = sector->sector_type;
  const bool has_effect = st == FLASH || st == FAST_STROBE_1 || st == FAST_STROBE_2 || st == FAST_STROBE_SYNC ||
                          st == SLOW_STROBE || st == SLOW_STROBE_SYNC || st == GLOW || st == FLICKER;
  if (!has_effect) return {base, std::nullopt};
  const float alt = light_to_f32(level.sector_min_light(sector));
  if (std::fabs(alt - base) < F32_EPSILON) return {base, std::nullopt};
  const float sync =
      (st == SLOW_STROBE_SYNC || st == FAST_STROBE_SYNC || st == GLOW) ? 0.0f : id_to_sync(level.sector_id(sector));
  LightEffect e{alt, 0.0f, 0.0f, sync, LightEffectKind::Glow};
  switch (st) {
    case FLASH: e.kind = LightEffectKind::Random, e.speed = 20.0f, e.duration = 0.06f; break;
    case FLICKER: e.kind = LightEffectKind::Random, e.speed = 8.0f, e.duration = 0.5f; break;
    case SLOW_STROBE:
    case SLOW_STROBE_SYNC: e.kind = LightEffectKind::Alternate, e.speed = 1.0f, e.duration = 0.85f; break;
    case FAST_STROBE_1:
    case FAST_STROBE_2:
    case FAST_STROBE_SYNC: e.kind = LightEffectKind::Alternate, e.speed = 2.0f, e.duration = 0.7f; break;
    default: e.kind = LightEffectKind::Glow, e.speed = 0.5f, e.duration = 0.0f; break;
  }
  return {base, e};
}

LightInfo with_contrast(const LightInfo &info, Contrast c) {
  const float contrast = c == Contrast::Darken ? -2.0f / 31.0f : 2.0f / 31.0f;
  return {clamp01(info.level + contrast), info.effect};
}

// ---- math/src/line.rs ----------------------------------------------------------------------------
Line2f Line2f::from_two_points(Pnt2f origin, Pnt2f towards) {
  const float dx = towards.x - origin.x, dy = towards.y - origin.y;
  const float length = magnitude(dx, dy);
  if (std::fabs(length) >= 1e-16f) return {origin, {dx / length, dy / length}, length};
  return {origin, {0.0f, 0.0f}, 0.0f};
}

std::optional<Pnt2f> Line2f::intersect_point(const Line2f &o) const {
  const float den = displace.x * o.displace.y - displace.y * o.displace.x;
  if (std::fabs(den) < 1e-16f) return std::nullopt;
  const float ex = o.origin.x - origin.x, ey = o.origin.y - origin.y;
  const float off = (ex * o.displace.y - ey * o.displace.x) / den;
  return Pnt2f{origin.x + displace.x * off, origin.y + displace.y * off};
}

Line2f partition_line(const WadNode &n) {
  return Line2f::from_two_points(from_wad_coords(n.line_x, n.line_y),
                                 from_wad_coords((int16_t)(n.line_x + n.step_x), (int16_t)(n.line_y + n.step_y)));
}

// ---- LevelAnalysis (visitor.rs:316-497) ------------------------------------------------------------
namespace {
std::optional<int16_t> to_height(const HeightDef &d, const WadSector &s, const NeighbourHeights &h) {  // :273-286
  int16_t base;
  switch (d.to) {
    case HeightRef::LowestFloor: base = h.lowest_floor; break;
    case HeightRef::NextFloor:
      if (!h.next_floor) return std::nullopt;
      base = *h.next_floor;
      break;
    case HeightRef::HighestFloor: base = h.highest_floor; break;
    case HeightRef::LowestCeiling: base = h.lowest_ceiling; break;
    case HeightRef::HighestCeiling: base = h.highest_ceiling; break;
    case HeightRef::Floor: base = s.floor_height; break;
    default: base = s.ceiling_height; break;
  }
  return (int16_t)(base + d.offset);
}

void merge_range(std::optional<std::pair<int16_t, int16_t>> &range, int16_t current,
                 const std::optional<int16_t> &a, const std::optional<int16_t> &b) {  // :247-261
  for (const auto &c : {a, b}) {
    if (!c) continue;
    if (range)
      range = std::make_pair(std::min(range->first, *c), std::max(range->second, *c));
    else
      range = std::make_pair(*c, *c);
  }
  if (range) range = std::make_pair(std::min(range->first, current), std::max(range->second, current));
}

void update_dynamic(DynamicSectorInfo &info, uint32_t &next_id, const Level &level, uint16_t sector_id,
                    const std::optional<MoveEffectDef> &move) {  // DynamicSectorInfo::update :168-244
  if (!move) return;
  const WadSector &sector = level.sectors[sector_id];
  if (!info.neighbour_heights) {
    auto h = level.neighbour_heights(&sector);
    if (!h) return;  // "Sector has no neighbours"
    info.neighbour_heights = h;
  }
  const NeighbourHeights &h = *info.neighbour_heights;
  std::optional<int16_t> ff, sf, fc, sc;
  if (move->floor) {
    ff = to_height(move->floor->first, sector, h);
    if (move->floor->second) sf = to_height(*move->floor->second, sector, h);
  }
  if (move->ceiling) {
    fc = to_height(move->ceiling->first, sector, h);
    if (move->ceiling->second) sc = to_height(*move->ceiling->second, sector, h);
  }
  merge_range(info.floor_range, sector.floor_height, ff, sf);
  merge_range(info.ceiling_range, sector.ceiling_height, fc, sc);
  if (info.ceiling_range && info.ceiling_id.v == 0) info.ceiling_id.v = next_id++;
  if (info.floor_range && info.floor_id.v == 0) info.floor_id.v = next_id++;
}
}  // namespace

LevelAnalysis::LevelAnalysis(const Level &level, const WadMetadata &meta) {
  std::vector<std::pair<uint16_t, uint16_t>> tags;  // (tag, sector id), sorted
  for (size_t i = 0; i < level.sectors.size(); i++)
    if (level.sectors[i].tag > 0) tags.emplace_back(level.sectors[i].tag, (uint16_t)i);
  std::sort(tags.begin(), tags.end());
  if (tags.empty()) return;  // visitor.rs:360-364 (num_objects stays 0; callers use max(1, ..))
  uint32_t next_id = 1;
  for (const WadLinedef &ld : level.linedefs) {
    if (ld.special_type == 0) continue;
    if (!level.vertex(ld.start_vertex) || !level.vertex(ld.end_vertex)) continue;
    auto it = meta.linedef.find(ld.special_type);
    const std::optional<MoveEffectDef> none;
    const std::optional<MoveEffectDef> &move = it != meta.linedef.end() ? it->second.move_effect : none;
    num_triggers_++;
    if (ld.sector_tag == 0) {  // manual linedef: acts on its left sector (visitor.rs:385-404)
      // (a left side naming a sector that does not exist: the reference indexes level.sectors[id], visitor.rs:175 -- a
      // panic; defined here as the warn-and-skip the walker applies to every other dangling reference)
      if (const WadSidedef *left = level.side(ld.left_side))
        if (left->sector < level.sectors.size())
          update_dynamic(dynamic_info_[left->sector], next_id, level, left->sector, move);
      continue;
    }
    auto first = std::lower_bound(tags.begin(), tags.end(), std::make_pair(ld.sector_tag, (uint16_t)0));
    for (; first != tags.end() && first->first == ld.sector_tag; ++first)
      update_dynamic(dynamic_info_[first->second], next_id, level, first->second, move);
  }
  num_objects_ = next_id;
}

// ---- points_to_polygon (visitor.rs:1184-1259) -------------------------------------------------------
namespace {
Pnt2f polygon_center(const std::vector<Pnt2f> &pts) {
  float cx = 0.0f, cy = 0.0f;
  for (const Pnt2f &p : pts) {
    cx += p.x;
    cy += p.y;
  }
  const float n = (float)pts.size();
  return {cx / n, cy / n};
}

// true iff the reference comparator (visitor.rs:1195-1224) returns Ordering::Less for (a, b)
bool poly_less(const Pnt2f &a, const Pnt2f &b, const Pnt2f &c) {
  const float acx = a.x - c.x, acy = a.y - c.y, bcx = b.x - c.x, bcy = b.y - c.y;
  if (acx >= 0.0f && bcx < 0.0f) return true;
  if (acx < 0.0f && bcx >= 0.0f) return false;
  if (acx == 0.0f && bcx == 0.0f) {
    if (acy >= 0.0f || bcy >= 0.0f) return a.y > b.y;
    return b.y > a.y;
  }
  return acx * bcy - acy * bcx < 0.0f;
}
}  // namespace

void points_to_polygon(std::vector<Pnt2f> &points) {
  if (points.size() < 2) {
    points.clear();
    return;
  }
  const Pnt2f c = polygon_center(points);
  // The comparator never answers Equal and is not a strict weak order, so std::sort would be
  // undefined behaviour.  Pinned algorithm (DESIGN.md "polygon sort"): linear insertion sort; it
  // agrees with Rust's sort_unstable_by whenever no two points are comparator-ambiguous.
  for (size_t i = 1; i < points.size(); i++)
    for (size_t j = i; j > 0 && poly_less(points[j], points[j - 1], c); j--) std::swap(points[j], points[j - 1]);
  std::vector<Pnt2f> simplified;
  simplified.reserve(points.size());
  simplified.push_back(points[0]);
  Pnt2f current = points[1];
  float area = 0.0f;
  for (size_t i = 2; i < points.size(); i++) {
    const Pnt2f next = points[i], prev = simplified.back();
    const float new_area =
        ((next.x - current.x) * (current.y - prev.y) - (next.y - current.y) * (current.x - prev.x)) * 0.5f;
    if (new_area >= 0.0f) {
      if (area + new_area > 1.024e-5f) {
        area = 0.0f;
        simplified.push_back(current);
      } else {
        area += new_area;
      }
    }
    current = next;
  }
  simplified.push_back(points.back());
  if (simplified.size() < 3) {
    points.clear();
    return;
  }
  while (simplified.size() > 1 &&
         magnitude(simplified[0].x - simplified.back().x, simplified[0].y - simplified.back().y) < 0.0032f)
    simplified.pop_back();
  const Pnt2f center = polygon_center(simplified);
  for (Pnt2f &p : simplified) {
    const Pnt2f n = normalize_or_zero(p.x - center.x, p.y - center.y);
    p.x += n.x * POLY_BIAS;
    p.y += n.y * POLY_BIAS;
  }
  points = std::move(simplified);
}

// ---- LevelWalker -----------------------------------------------------------------------------------
LevelWalker::LevelWalker(const Level &level, const LevelAnalysis &analysis, const TextureDirectory &tex,
                         const WadMetadata &meta, LevelVisitor &visitor)
    : level_(level), analysis_(analysis), tex_(tex), meta_(meta), visitor_(visitor) {
  int mn = 32767, mx = -32768;  // min_max_height (visitor.rs:1173-1182)
  for (const WadSector &s : level.sectors) {
    mn = std::min<int>(mn, s.floor_height);
    mx = std::max<int>(mx, s.ceiling_height);
  }
  height_range_ = {(int16_t)(mn - 512), (int16_t)(mx + 512)};
  bsp_lines_.reserve(32);
}

void LevelWalker::walk() {
  if (level_.nodes.empty()) return;  // "Level contains no nodes, visitor not called at all."
  const WadNode &root = level_.nodes.back();
  const Line2f partition = partition_line(root);
  visitor_.visit_bsp_root(partition);
  children(root, partition);
  visitor_.visit_bsp_node_end();
  things();
}

SectorInfo LevelWalker::sector_info(const WadSector *s) const {  // visitor.rs:569-588
  SectorInfo info{{0}, {0}, {s->floor_height, s->floor_height}, {s->ceiling_height, s->ceiling_height}};
  if (const DynamicSectorInfo *d = analysis_.dynamic(level_.sector_id(s))) {
    info.floor_id = d->floor_id;
    info.ceiling_id = d->ceiling_id;
    if (d->floor_range) info.floor_range = *d->floor_range;
    if (d->ceiling_range) info.ceiling_range = *d->ceiling_range;
  }
  return info;
}

const LightInfo *LevelWalker::light_info(const WadSector *s) {  // visitor.rs:1140-1148
  const uint16_t id = level_.sector_id(s);
  auto it = light_cache_.find(id);
  if (it == light_cache_.end()) it = light_cache_.emplace(id, new_light(level_, s)).first;
  return &it->second;
}

void LevelWalker::node(uint16_t id, Branch branch) {  // visitor.rs:590-609
  const size_t idx = id & 0x7FFF;
  if (id & 0x8000) {
    visitor_.visit_bsp_leaf(branch);
    subsector(idx);
    visitor_.visit_bsp_leaf_end();
    return;
  }
  if (idx >= level_.nodes.size()) return;  // "Missing entire node"
  const WadNode &n = level_.nodes[idx];
  const Line2f partition = partition_line(n);
  visitor_.visit_bsp_node(partition, branch);
  children(n, partition);
  visitor_.visit_bsp_node_end();
}

void LevelWalker::children(const WadNode &n, const Line2f &partition) {  // visitor.rs:611-619
  bsp_lines_.push_back(partition);
  node(n.left, Branch::Positive);
  bsp_lines_.pop_back();
  bsp_lines_.push_back(partition.inverted_halfspaces());
  node(n.right, Branch::Negative);
  bsp_lines_.pop_back();
}

void LevelWalker::subsector(size_t id) {  // visitor.rs:621-709
  if (id >= level_.subsectors.size()) return;
  const WadSubsector ss = level_.subsectors[id];
  if ((size_t)ss.first_seg + ss.num_segs > level_.segs.size()) return;
  if (ss.num_segs == 0) return;
  const WadSeg *segs = &level_.segs[ss.first_seg];
  const WadSector *sector = level_.seg_sector(segs[0]);
  if (!sector) return;
  const SectorInfo info = sector_info(sector);
  subsector_seg_lines_.clear();
  subsector_points_.clear();
  for (size_t i = 0; i < ss.num_segs; i++) {
    const auto v1 = level_.vertex(segs[i].start_vertex), v2 = level_.vertex(segs[i].end_vertex);
    if (!v1 || !v2) return;
    subsector_points_.push_back(*v1);
    subsector_points_.push_back(*v2);
    subsector_seg_lines_.push_back(Line2f::from_two_points(*v1, *v2));
    seg(sector, info, segs[i], *v1, *v2);
  }
  if (record_leaves) record_leaves->push_back({(uint32_t)id, bsp_lines_});
  if (precomputed_polygons) {
    subsector_points_ = (*precomputed_polygons)[id];
  } else {
    // implicit points: pairwise intersections of the BSP half-plane stack that lie inside every BSP
    // half-plane and on the inner side of every seg (visitor.rs:672-691)
    const size_t nb = bsp_lines_.size();
    for (size_t i = 0; i + 1 < nb; i++) {
      for (size_t j = i + 1; j < nb; j++) {
        const auto p = bsp_lines_[i].intersect_point(bsp_lines_[j]);
        if (!p) continue;
        bool inside = true;
        for (size_t k = 0; k < nb && inside; k++) inside = bsp_lines_[k].signed_distance(*p) >= -BSP_TOLERANCE;
        for (size_t k = 0; k < subsector_seg_lines_.size() && inside; k++)
          inside = subsector_seg_lines_[k].signed_distance(*p) <= SEG_TOLERANCE;
        if (inside) subsector_points_.push_back(*p);
      }
    }
    points_to_polygon(subsector_points_);
  }
  if (subsector_points_.size() >= 3) flat_poly(sector, info);
}

void LevelWalker::seg(const WadSector *sector, const SectorInfo &info, const WadSeg &sg, Pnt2f v1,
                      Pnt2f v2) {  // visitor.rs:711-837
  if (record_segs || precomputed_segs) {
    const size_t index = (size_t)(&sg - level_.segs.data());
    if (record_segs) {
      SegInput in{};
      if (seg_input(sector, info, sg, v1, v2, in)) (*record_segs)[index] = in;
      return;
    }
    emit_seg_geometry(sector, sg, (*precomputed_segs)[index]);
    return;
  }
  const WadLinedef *line = level_.seg_linedef(sg);
  if (!line) return;
  const WadSidedef *sidedef = level_.seg_sidedef(sg);
  if (!sidedef) return;
  const int16_t mn = height_range_.first, mx = height_range_.second;
  const int16_t floor = sector->floor_height, ceiling = sector->ceiling_height;
  const bool unpeg_lower = line->lower_unpegged();
  const WadSector *back = level_.seg_back_sector(sg);
  if (!back) {
    InternalWallQuad q{unpeg_lower ? info.floor_id : info.ceiling_id,
                       sector,
                       &sg,
                       v1,
                       v2,
                       unpeg_lower ? floor : (int16_t)(ceiling - info.max_height()),
                       unpeg_lower ? (int16_t)(floor + info.max_height()) : ceiling,
                       sidedef->middle_texture,
                       unpeg_lower ? Peg::Bottom : Peg::Top,
                       true};
    wall_quad(q);
    if (sector->ceiling_texture.is_sky_flat()) sky_quad(info.ceiling_id, v1, v2, ceiling, mx);
    if (sector->floor_texture.is_sky_flat()) sky_quad(info.floor_id, v1, v2, mn, floor);
    return;
  }
  const int16_t back_floor = back->floor_height, back_ceiling = back->ceiling_height;
  const SectorInfo back_info = sector_info(back);
  if (sector->ceiling_texture.is_sky_flat() && !back->ceiling_texture.is_sky_flat())
    sky_quad(info.ceiling_id, v1, v2, ceiling, mx);
  if (sector->floor_texture.is_sky_flat() && !back->floor_texture.is_sky_flat())
    sky_quad(info.floor_id, v1, v2, mn, floor);
  const bool unpeg_upper = line->upper_unpegged();
  int16_t fl, ce;
  if (back_info.floor_range.second > info.floor_range.first) {
    wall_quad({back_info.floor_id, sector, &sg, v1, v2,
               (int16_t)(back_floor - back_info.floor_range.second + info.floor_range.first), back_floor,
               sidedef->lower_texture, unpeg_lower ? Peg::BottomLower : Peg::Top, true});
    fl = back_floor;
  } else {
    fl = floor;
  }
  if (back_ceiling < ceiling) {
    if (!back->ceiling_texture.is_sky_flat())
      wall_quad({back_info.ceiling_id, sector, &sg, v1, v2, back_ceiling, ceiling, sidedef->upper_texture,
                 unpeg_upper ? Peg::Top : Peg::Bottom, true});
    ce = back_ceiling;
  } else {
    ce = ceiling;
  }
  Peg peg;
  if (unpeg_lower)
    peg = sidedef->upper_texture.is_untextured() ? Peg::TopFloat : Peg::Bottom;
  else
    peg = sidedef->lower_texture.is_untextured() ? Peg::BottomFloat : Peg::Top;
  wall_quad({unpeg_lower ? info.floor_id : info.ceiling_id, sector, &sg, v1, v2, fl, ce, sidedef->middle_texture, peg,
             line->impassable()});
}

// Inputs of the device SEG kernel: every lookup seg() / wall_quad() / sky_quad() perform, resolved.
bool LevelWalker::seg_input(const WadSector *sector, const SectorInfo &info, const WadSeg &sg, Pnt2f v1, Pnt2f v2,
                            SegInput &in) {
  const WadLinedef *line = level_.seg_linedef(sg);
  const WadSidedef *sidedef = level_.seg_sidedef(sg);
  if (!line || !sidedef) return false;
  in.v1x = v1.x, in.v1y = v1.y, in.v2x = v2.x, in.v2y = v2.y;
  in.floor = sector->floor_height, in.ceiling = sector->ceiling_height;
  in.f_floor_lo = info.floor_range.first, in.f_floor_hi = info.floor_range.second;
  in.f_ceil_lo = info.ceiling_range.first, in.f_ceil_hi = info.ceiling_range.second;
  in.mn = height_range_.first, in.mx = height_range_.second;
  in.x_offset = sidedef->x_offset, in.y_offset = sidedef->y_offset;
  in.seg_offset = sg.offset;
  const WadName names[3] = {sidedef->lower_texture, sidedef->upper_texture, sidedef->middle_texture};
  for (int k = 0; k < 3; k++) {
    if (names[k].is_untextured()) {
      in.tex_h[k] = -1;
    } else {
      const Image *image = tex_.texture(names[k]);
      in.tex_h[k] = image ? (int16_t)image->height() : (int16_t)-2;
    }
  }
  in.f_floor_id = info.floor_id.v, in.f_ceil_id = info.ceiling_id.v;
  uint32_t flags = SEG_VALID;
  if (line->lower_unpegged()) flags |= SEG_UNPEG_LOWER;
  if (line->upper_unpegged()) flags |= SEG_UNPEG_UPPER;
  if (line->impassable()) flags |= SEG_IMPASSABLE;
  if (line->special_type == 0x30) flags |= SEG_SCROLL;
  if (sector->ceiling_texture.is_sky_flat()) flags |= SEG_F_CEIL_SKY;
  if (sector->floor_texture.is_sky_flat()) flags |= SEG_F_FLOOR_SKY;
  if (light_info(sector)->effect) flags |= SEG_LIGHT_EFFECT;
  if (const WadSector *back = level_.seg_back_sector(sg)) {
    flags |= SEG_HAS_BACK;
    const SectorInfo bi = sector_info(back);
    in.back_floor = back->floor_height, in.back_ceiling = back->ceiling_height;
    in.b_floor_lo = bi.floor_range.first, in.b_floor_hi = bi.floor_range.second;
    in.b_ceil_lo = bi.ceiling_range.first, in.b_ceil_hi = bi.ceiling_range.second;
    in.b_floor_id = bi.floor_id.v, in.b_ceil_id = bi.ceiling_id.v;
    if (back->ceiling_texture.is_sky_flat()) flags |= SEG_B_CEIL_SKY;
    if (back->floor_texture.is_sky_flat()) flags |= SEG_B_FLOOR_SKY;
  }
  in.flags = flags;
  return true;
}

// Visitor events from the device's results, in the order seg() emits them (visitor.rs:711-837).
void LevelWalker::emit_seg_geometry(const WadSector *sector, const WadSeg &sg, const SegGeometry &g) {
  const WadSidedef *sidedef = level_.seg_sidedef(sg);
  if (!sidedef) return;
  const WadName names[3] = {sidedef->lower_texture, sidedef->upper_texture, sidedef->middle_texture};
  auto quad = [&](const SegQuadGeometry &q) {
    if (!q.valid) return;
    const LightInfo *light = light_info(sector);
    LightInfo contrasted;
    if (q.contrast) {
      contrasted = with_contrast(*light, q.contrast == 1 ? Contrast::Brighten : Contrast::Darken);
      light = &contrasted;
    }
    StaticQuad out;
    out.object_id = ObjectId{q.object_id};
    out.v1 = {q.v1x, q.v1y};
    out.v2 = {q.v2x, q.v2y};
    out.tex_start[0] = q.s1, out.tex_start[1] = q.t1;
    out.tex_end[0] = q.s2, out.tex_end[1] = q.t2;
    out.height_range[0] = q.low, out.height_range[1] = q.high;
    out.light_info = light;
    out.scroll = q.scroll;
    if (!names[q.slot].is_untextured()) out.tex_name = names[q.slot];
    out.blocker = q.blocker != 0;
    visitor_.visit_wall_quad(out);
  };
  auto sky = [&](const SegSkyGeometry &k) {
    if (!k.valid) return;
    SkyQuad q;
    q.object_id = ObjectId{k.object_id};
    q.v1 = {k.v1x, k.v1y};
    q.v2 = {k.v2x, k.v2y};
    q.height_range[0] = k.low, q.height_range[1] = k.high;
    visitor_.visit_sky_quad(q);
  };
  if (!level_.seg_back_sector(sg)) {  // one-sided: wall, then sky above / below
    quad(g.quad[0]);
    sky(g.sky[0]);
    sky(g.sky[1]);
  } else {  // two-sided: sky, then lower, upper, middle
    sky(g.sky[0]);
    sky(g.sky[1]);
    quad(g.quad[0]);
    quad(g.quad[1]);
    quad(g.quad[2]);
  }
}

void LevelWalker::wall_quad(const InternalWallQuad &q) {  // visitor.rs:839-937
  if (q.low >= q.high) return;
  std::optional<std::pair<float, float>> size;
  if (!q.texture_name.is_untextured()) {
    const Image *image = tex_.texture(q.texture_name);
    if (!image) return;  // "wall_quad: No such wall texture"
    size = std::make_pair((float)image->width(), (float)image->height());
  }
  const WadLinedef *line = level_.seg_linedef(*q.seg);
  const WadSidedef *sidedef = level_.seg_sidedef(*q.seg);
  if (!line || !sidedef) return;
  const Pnt2f dir = normalize_or_zero(q.v2.x - q.v1.x, q.v2.y - q.v1.y);
  const float bx = dir.x * POLY_BIAS, by = dir.y * POLY_BIAS;
  const Pnt2f v1{q.v1.x + (-bx), q.v1.y + (-by)}, v2{q.v2.x + bx, q.v2.y + by};
  float low, high;
  if (size && q.peg == Peg::TopFloat) {
    low = from_wad_height((int16_t)(q.low + sidedef->y_offset));
    high = from_wad_height((int16_t)(q.low + (int16_t)size->second + sidedef->y_offset));
  } else if (size && q.peg == Peg::BottomFloat) {
    low = from_wad_height((int16_t)(q.high + sidedef->y_offset - (int16_t)size->second));
    high = from_wad_height((int16_t)(q.high + sidedef->y_offset));
  } else {
    low = from_wad_height(q.low);
    high = from_wad_height(q.high);
  }
  const LightInfo *light = light_info(q.sector);
  LightInfo contrasted;
  if (!light->effect) {  // fake contrast (visitor.rs:889-901)
    if (std::fabs(v1.x - v2.x) < F32_EPSILON) {
      contrasted = with_contrast(*light, Contrast::Brighten);
      light = &contrasted;
    } else if (std::fabs(v1.y - v2.y) < F32_EPSILON) {
      contrasted = with_contrast(*light, Contrast::Darken);
      light = &contrasted;
    }
  }
  const float height = to_wad_height(high - low);
  const float s1 = (float)q.seg->offset + (float)sidedef->x_offset;
  const float s2 = s1 + to_wad_height(magnitude(v2.x - v1.x, v2.y - v1.y));
  float t1, t2;
  if (!size || q.peg == Peg::Top) {
    t1 = height;
    t2 = 0.0f;
  } else if (q.peg == Peg::Bottom) {
    t1 = size->second;
    t2 = size->second - height;
  } else if (q.peg == Peg::BottomLower) {
    const float sector_height = (float)(int16_t)(q.sector->ceiling_height - q.sector->floor_height);
    t1 = size->second + sector_height;
    t2 = size->second - height + sector_height;
  } else {
    t1 = size->second;
    t2 = 0.0f;
  }
  t1 = t1 + (float)sidedef->y_offset;
  t2 = t2 + (float)sidedef->y_offset;
  StaticQuad out;
  out.object_id = q.object_id;
  out.v1 = v1;
  out.v2 = v2;
  out.tex_start[0] = s1;
  out.tex_start[1] = t1;
  out.tex_end[0] = s2;
  out.tex_end[1] = t2;
  out.height_range[0] = low - POLY_BIAS;
  out.height_range[1] = high + POLY_BIAS;
  out.light_info = light;
  out.scroll = line->special_type == 0x30 ? 35.0f : 0.0f;
  if (size) out.tex_name = q.texture_name;
  out.blocker = q.blocker;
  visitor_.visit_wall_quad(out);
}

void LevelWalker::flat_poly(const WadSector *sector, const SectorInfo &info) {  // visitor.rs:939-985
  const LightInfo *light = light_info(sector);
  const bool floor_sky = sector->floor_texture.is_sky_flat(), ceil_sky = sector->ceiling_texture.is_sky_flat();
  const float floor_y = from_wad_height(floor_sky ? height_range_.first : sector->floor_height);
  const float ceil_y = from_wad_height(ceil_sky ? height_range_.second : sector->ceiling_height);
  const Pnt2f *pts = subsector_points_.data();
  const size_t n = subsector_points_.size();
  if (floor_sky)
    visitor_.visit_floor_sky_poly({info.floor_id, pts, n, floor_y});
  else
    visitor_.visit_floor_poly({info.floor_id, pts, n, floor_y, light, sector->floor_texture});
  if (ceil_sky)
    visitor_.visit_ceil_sky_poly({info.ceiling_id, pts, n, ceil_y});
  else
    visitor_.visit_ceil_poly({info.ceiling_id, pts, n, ceil_y, light, sector->ceiling_texture});
}

void LevelWalker::sky_quad(ObjectId id, Pnt2f v1, Pnt2f v2, int16_t low, int16_t high) {  // visitor.rs:987-1008
  if (low >= high) return;
  const Pnt2f edge = normalize_or_zero(v2.x - v1.x, v2.y - v1.y);
  const float bx = edge.x * POLY_BIAS * 16.0f, by = edge.y * POLY_BIAS * 16.0f;
  const float nx = -edge.y, ny = edge.x;
  const float nbx = nx * POLY_BIAS * 16.0f, nby = ny * POLY_BIAS * 16.0f;
  SkyQuad q;
  q.object_id = id;
  q.v1 = {v1.x + (nbx - bx), v1.y + (nby - by)};
  q.v2 = {v2.x + (nbx + bx), v2.y + (nby + by)};
  q.height_range[0] = from_wad_height(low);
  q.height_range[1] = from_wad_height(high);
  visitor_.visit_sky_quad(q);
}

const WadSector *LevelWalker::sector_at(Pnt2f pos) const {  // visitor.rs:1028-1060
  uint16_t child = (uint16_t)(level_.nodes.size() - 1);
  for (;;) {
    const size_t idx = child & 0x7FFF;
    if (child & 0x8000) {
      if (idx >= level_.subsectors.size()) return nullptr;
      const WadSubsector ss = level_.subsectors[idx];
      if ((size_t)ss.first_seg + ss.num_segs > level_.segs.size() || ss.num_segs == 0) return nullptr;
      const WadSeg *segs = &level_.segs[ss.first_seg];
      const WadSector *sector = level_.seg_sector(segs[0]);
      if (!sector) return nullptr;
      for (size_t i = 0; i < ss.num_segs; i++) {
        const auto a = level_.vertex(segs[i].start_vertex), b = level_.vertex(segs[i].end_vertex);
        if (!a || !b) continue;
        if (!(Line2f::from_two_points(*a, *b).signed_distance(pos) <= SEG_TOLERANCE)) return nullptr;
      }
      return sector;
    }
    if (idx >= level_.nodes.size()) return nullptr;
    const WadNode &n = level_.nodes[idx];
    child = partition_line(n).signed_distance(pos) > 0.0f ? n.left : n.right;
  }
}

void LevelWalker::things() {  // visitor.rs:1010-1026
  for (const WadThing &thing : level_.things) {
    const Pnt2f pos = from_wad_coords(thing.x, thing.y);
    const float yaw_deg = std::round((float)thing.angle / 45.0f) * 45.0f;  // f32::round: half away from zero
    const WadSector *sector = sector_at(pos);
    if (!sector) continue;
    std::optional<Marker> marker;
    switch (thing.thing_type) {
      case 1: marker = Marker{MarkerKind::StartPos, 0}; break;
      case 2: marker = Marker{MarkerKind::StartPos, 1}; break;
      case 3: marker = Marker{MarkerKind::StartPos, 2}; break;
      case 4: marker = Marker{MarkerKind::StartPos, 3}; break;
      case 11: marker = Marker{MarkerKind::TeleportStart, 0}; break;
      case 14: marker = Marker{MarkerKind::TeleportEnd, 0}; break;
      default: break;
    }
    if (marker) {
      const float p3[3] = {pos.x, from_wad_height(sector->floor_height), pos.y};
      visitor_.visit_marker(p3, yaw_deg * (float)(M_PI / 180.0), *marker);  // cgmath Rad::from(Deg)
    } else {
      decor(thing, pos, sector);
    }
  }
}

void LevelWalker::decor(const WadThing &thing, Pnt2f pos, const WadSector *sector) {  // visitor.rs:1062-1137
  const ThingMetadata *meta = meta_.find_thing(thing.thing_type);
  if (!meta) return;
  WadName sprite0 = meta->sprite;
  if (!meta->sequence.empty()) (void)sprite0.push((uint8_t)meta->sequence[0]);
  WadName sprite1 = sprite0;
  if (!sprite0.push('0') || !sprite1.push('1')) return;
  WadName name = sprite0;
  const Image *image = tex_.texture(sprite0);
  if (!image) {
    image = tex_.texture(sprite1);
    name = sprite1;
    if (!image) return;  // "No such sprite"
  }
  const float sx = from_wad_height((int16_t)image->width()), sy = from_wad_height((int16_t)image->height());
  const DynamicSectorInfo *dyn = analysis_.dynamic(level_.sector_id(sector));
  Decor d;
  if (meta->hanging) {
    d.object_id = dyn ? dyn->ceiling_id : ObjectId{0};
    const float top = from_wad_height(sector->ceiling_height);
    d.low[0] = pos.x, d.low[1] = top - sy, d.low[2] = pos.y;
    d.high[0] = pos.x, d.high[1] = top, d.high[2] = pos.y;
  } else {
    d.object_id = dyn ? dyn->floor_id : ObjectId{0};
    const float bottom = from_wad_height(sector->floor_height);
    d.low[0] = pos.x, d.low[1] = bottom, d.low[2] = pos.y;
    d.high[0] = pos.x, d.high[1] = bottom + sy, d.high[2] = pos.y;
  }
  d.half_width = sx * 0.5f;
  d.light_info = light_info(sector);
  d.tex_name = name;
  visitor_.visit_decor(d);
}

}  // namespace rdoom::wad
