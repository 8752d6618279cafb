// wad::{name, archive, level}: IWAD directory, typed lump decoding, level records and navigation.
// Reference: wad/src/name.rs, wad/src/archive.rs, wad/src/level.rs, wad/src/types.rs.
#include <algorithm>
#include <cstring>
#include <fstream>

#include "wad.hpp"

namespace rdoom::wad {

namespace {
inline uint16_t rd_u16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline int16_t rd_i16(const uint8_t *p) { return (int16_t)rd_u16(p); }
inline int32_t rd_i32(const uint8_t *p) {
  return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}
}  // namespace

// ---- WadName -----------------------------------------------------------------------------------
bool WadName::valid_byte(uint8_t b) {
  return (b >= 'A' && b <= 'Z') || (b >= '0' && b <= '9') || b == '_' || b == '%' || b == '-' || b == '[' ||
         b == ']' || b == '\\';
}

WadName WadName::from_bytes(const uint8_t *value, size_t len) {
  WadName out;
  bool nulled = false;
  for (size_t i = 0; i < len && i < 8; i++) {
    const uint8_t src = value[i];
    if (src >= 0x80) throw WadError(RDOOM_CORRUPT_WAD, "invalid byte in wad name");
    const uint8_t up = (src >= 'a' && src <= 'z') ? (uint8_t)(src - 32) : src;
    if (up == 0) {
      nulled = true;
      break;
    }
    if (!valid_byte(up)) throw WadError(RDOOM_CORRUPT_WAD, "invalid byte in wad name");
    out.b[i] = up;
  }
  if (!(nulled || len <= 8)) throw WadError(RDOOM_CORRUPT_WAD, "wad name too long");
  return out;
}

bool WadName::push(uint8_t byte) {
  const uint8_t up = (byte >= 'a' && byte <= 'z') ? (uint8_t)(byte - 32) : byte;
  if (!valid_byte(up)) return false;
  for (auto &c : b)
    if (c == 0) {
      c = up;
      return true;
    }
  return false;
}

std::string WadName::str() const {
  std::string s;
  for (auto c : b) {
    if (!c) break;
    s.push_back((char)c);
  }
  return s;
}

// ---- Archive -----------------------------------------------------------------------------------
std::unique_ptr<Archive> Archive::open(const std::string &wad_path, const std::string &meta_path) {
  auto a = std::make_unique<Archive>();
  {
    std::ifstream f(wad_path, std::ios::binary | std::ios::ate);
    if (!f) throw WadError(RDOOM_IO, "cannot open wad file '" + wad_path + "'");
    const std::streamsize n = f.tellg();
    f.seekg(0);
    a->data_.resize((size_t)n);
    if (n > 0 && !f.read((char *)a->data_.data(), n)) throw WadError(RDOOM_IO, "cannot read wad file");
  }
  const auto &d = a->data_;
  if (d.size() < 12) throw WadError(RDOOM_CORRUPT_WAD, "bad wad header");
  if (std::memcmp(d.data(), "IWAD", 4) != 0) throw WadError(RDOOM_CORRUPT_WAD, "bad wad header identifier");
  const int32_t num_lumps = rd_i32(&d[4]), table = rd_i32(&d[8]);
  if (num_lumps < 0 || table < 0 || (uint64_t)table + 16ull * (uint64_t)num_lumps > d.size())
    throw WadError(RDOOM_CORRUPT_WAD, "bad lump info table");
  for (int32_t i = 0; i < num_lumps; i++) {
    const uint8_t *e = &d[(size_t)table + 16u * (size_t)i];
    LumpInfo li;
    li.name = WadName::from_bytes(e + 8, 8);
    li.offset = (uint64_t)(uint32_t)rd_i32(e);
    li.size = (size_t)(uint32_t)rd_i32(e + 4);
    li.outside = li.size > 0 && li.offset + li.size > d.size();  // reported when the lump is read (lump_data)
    if (li.size == 0 || li.outside) li.offset = 0;
    a->index_map_[li.name] = a->lumps_.size();  // last duplicate wins (archive.rs:85)
    a->lumps_.push_back(li);
    if (li.name == WadName::from_str("THINGS")) {
      if (i == 0) throw WadError(RDOOM_CORRUPT_WAD, "THINGS lump without a level marker");
      a->levels_.push_back((size_t)i - 1);
    }
  }
  a->meta_ = WadMetadata::from_file(meta_path);
  return a;
}

size_t Archive::level_lump_index(size_t level) const {
  if (level >= levels_.size()) throw WadError(RDOOM_BAD_ARG, "level index out of range");
  return levels_[level];
}

const uint8_t *Archive::lump_data(size_t index) const {
  const LumpInfo &li = lump(index);
  if (li.outside) throw WadError(RDOOM_IO, "reading lump " + li.name.str() + " failed: it lies outside the file");
  return data_.data() + li.offset;
}

const LumpInfo &Archive::lump(size_t index) const {
  if (index >= lumps_.size()) throw WadError(RDOOM_CORRUPT_WAD, "missing required lump index");
  return lumps_[index];
}

std::optional<size_t> Archive::named_lump(const WadName &n) const {
  auto it = index_map_.find(n);
  if (it == index_map_.end()) return std::nullopt;
  return it->second;
}

size_t Archive::required_named_lump(const char *name) const {
  auto i = named_lump(WadName::from_str(name));
  if (!i) throw WadError(RDOOM_CORRUPT_WAD, std::string("missing required lump ") + name);
  return *i;
}

size_t Archive::checked_count(size_t index, size_t record) const {
  const LumpInfo &li = lump(index);
  if (!(li.size > 0 && li.size % record == 0))
    throw WadError(RDOOM_CORRUPT_WAD, "bad lump size for " + li.name.str());
  return li.size / record;
}

// ---- Level -------------------------------------------------------------------------------------
Level Level::from_archive(const Archive &wad, size_t index) {
  Level L;
  const size_t s = wad.level_lump_index(index);
  L.name = wad.lump(s).name;
  size_t n;
  const uint8_t *p;
  n = wad.checked_count(s + 1, 10);
  p = wad.lump_data(s + 1);
  for (size_t i = 0; i < n; i++, p += 10)
    L.things.push_back({rd_i16(p), rd_i16(p + 2), rd_i16(p + 4), rd_u16(p + 6), rd_u16(p + 8)});
  n = wad.checked_count(s + 2, 14);
  p = wad.lump_data(s + 2);
  for (size_t i = 0; i < n; i++, p += 14)
    L.linedefs.push_back({rd_u16(p), rd_u16(p + 2), rd_u16(p + 4), rd_u16(p + 6), rd_u16(p + 8), rd_i16(p + 10),
                          rd_i16(p + 12)});
  n = wad.checked_count(s + 4, 4);
  p = wad.lump_data(s + 4);
  for (size_t i = 0; i < n; i++, p += 4) L.vertices.push_back({rd_i16(p), rd_i16(p + 2)});
  n = wad.checked_count(s + 5, 12);
  p = wad.lump_data(s + 5);
  for (size_t i = 0; i < n; i++, p += 12)
    L.segs.push_back({rd_u16(p), rd_u16(p + 2), rd_u16(p + 4), rd_u16(p + 6), rd_u16(p + 8), rd_u16(p + 10)});
  n = wad.checked_count(s + 6, 4);
  p = wad.lump_data(s + 6);
  for (size_t i = 0; i < n; i++, p += 4) L.subsectors.push_back({rd_u16(p), rd_u16(p + 2)});
  n = wad.checked_count(s + 7, 28);
  p = wad.lump_data(s + 7);
  for (size_t i = 0; i < n; i++, p += 28) {
    WadNode nd;
    nd.line_x = rd_i16(p);
    nd.line_y = rd_i16(p + 2);
    nd.step_x = rd_i16(p + 4);
    nd.step_y = rd_i16(p + 6);
    for (int k = 0; k < 8; k++) nd.bbox[k] = rd_i16(p + 8 + 2 * k);
    nd.right = rd_u16(p + 24);
    nd.left = rd_u16(p + 26);
    L.nodes.push_back(nd);
  }
  n = wad.checked_count(s + 3, 30);
  p = wad.lump_data(s + 3);
  for (size_t i = 0; i < n; i++, p += 30) {
    WadSidedef sd;
    sd.x_offset = rd_i16(p);
    sd.y_offset = rd_i16(p + 2);
    sd.upper_texture = WadName::from_bytes(p + 4, 8);
    sd.lower_texture = WadName::from_bytes(p + 12, 8);
    sd.middle_texture = WadName::from_bytes(p + 20, 8);
    sd.sector = rd_u16(p + 28);
    L.sidedefs.push_back(sd);
  }
  n = wad.checked_count(s + 8, 26);
  p = wad.lump_data(s + 8);
  for (size_t i = 0; i < n; i++, p += 26) {
    WadSector sc;
    sc.floor_height = rd_i16(p);
    sc.ceiling_height = rd_i16(p + 2);
    sc.floor_texture = WadName::from_bytes(p + 4, 8);
    sc.ceiling_texture = WadName::from_bytes(p + 12, 8);
    sc.light = rd_i16(p + 20);
    sc.sector_type = rd_u16(p + 22);
    sc.tag = rd_u16(p + 24);
    L.sectors.push_back(sc);
  }
  return L;
}

std::optional<Pnt2f> Level::vertex(uint16_t id) const {
  if (id >= vertices.size()) return std::nullopt;
  return from_wad_coords(vertices[id].x, vertices[id].y);
}

const WadLinedef *Level::seg_linedef(const WadSeg &s) const {
  return s.linedef < linedefs.size() ? &linedefs[s.linedef] : nullptr;
}

const WadSidedef *Level::side(int16_t index) const {
  if (index < 0) return nullptr;  // -1 => None; other negatives index out of range (level.rs:139-151)
  return (size_t)index < sidedefs.size() ? &sidedefs[(size_t)index] : nullptr;
}

const WadSidedef *Level::seg_sidedef(const WadSeg &s) const {
  const WadLinedef *l = seg_linedef(s);
  if (!l) return nullptr;
  return s.direction == 0 ? side(l->right_side) : side(l->left_side);
}

const WadSidedef *Level::seg_back_sidedef(const WadSeg &s) const {
  const WadLinedef *l = seg_linedef(s);
  if (!l) return nullptr;
  return s.direction == 1 ? side(l->right_side) : side(l->left_side);
}

const WadSector *Level::sidedef_sector(const WadSidedef *s) const {
  if (!s) return nullptr;
  return s->sector < sectors.size() ? &sectors[s->sector] : nullptr;
}

template <class F>
void Level::for_adjacent_sectors(const WadSector *of, F f) const {
  const uint16_t id = sector_id(of);
  for (const WadLinedef &line : linedefs) {
    const WadSidedef *l = side(line.left_side);
    if (!l) continue;
    const WadSidedef *r = side(line.right_side);
    if (!r) continue;
    uint16_t adj;
    if (l->sector == id)
      adj = r->sector;
    else if (r->sector == id)
      adj = l->sector;
    else
      continue;
    if (adj < sectors.size()) f(sectors[adj]);
  }
}

int16_t Level::sector_min_light(const WadSector *of) const {
  int16_t m = of->light;
  for_adjacent_sectors(of, [&](const WadSector &s) { m = std::min(m, s.light); });
  return m;
}

std::optional<NeighbourHeights> Level::neighbour_heights(const WadSector *of) const {
  std::optional<NeighbourHeights> h;
  const int16_t of_floor = of->floor_height;
  for_adjacent_sectors(of, [&](const WadSector &s) {
    const int16_t floor = s.floor_height, ceil = s.ceiling_height;
    if (!h) {
      NeighbourHeights n{floor, floor, ceil, ceil, std::nullopt};
      if (floor > of_floor) n.next_floor = floor;
      h = n;
    } else {
      h->lowest_floor = std::min(h->lowest_floor, floor);
      h->highest_floor = std::max(h->highest_floor, floor);
      h->lowest_ceiling = std::min(h->lowest_ceiling, ceil);
      h->highest_ceiling = std::max(h->highest_ceiling, ceil);
      if (floor > of_floor) h->next_floor = h->next_floor ? std::min(*h->next_floor, floor) : floor;
    }
  });
  return h;
}

}  // namespace rdoom::wad
