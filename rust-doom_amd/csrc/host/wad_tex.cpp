// wad::{image, tex}: Doom picture decoding, TEXTURE1/2 composition, palette/colormap, atlases.
// Reference: wad/src/image.rs, wad/src/tex.rs.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "wad.hpp"

namespace rdoom::wad {
namespace {
inline uint16_t rd_u16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline int16_t rd_i16(const uint8_t *p) { return (int16_t)rd_u16(p); }
inline uint32_t rd_u32(const uint8_t *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
size_t next_pow2(size_t x) {  // tex.rs:348-354
  size_t p = 1;
  while (p < x) p *= 2;
  return p;
}
[[noreturn]] void corrupt(const std::string &why) { throw WadError(RDOOM_CORRUPT_WAD, why); }
}  // namespace

// ---- Image ---------------------------------------------------------------------------------------
Image::Image(size_t w, size_t h, uint16_t fill) : w_(w), h_(h) {
  if (w > MAX_IMAGE_SIZE || h > MAX_IMAGE_SIZE) corrupt("image too large");
  px_.assign(w * h, fill);
}

Image Image::from_buffer(const uint8_t *buf, size_t len) {
  if (len < 2) corrupt("Image missing width.");
  if (len < 4) corrupt("Image missing height.");
  const size_t w = rd_u16(buf), h = rd_u16(buf + 2);
  if (w > MAX_IMAGE_SIZE || h > MAX_IMAGE_SIZE) corrupt("image too large");
  if (len < 6) corrupt("Image missing x offset");
  if (len < 8) corrupt("Image missing y offset");
  Image img(w, h, 0xFFFF);  // `vec![!0; width * height]` (image.rs:63)
  img.x_offset = rd_i16(buf + 4);
  img.y_offset = rd_i16(buf + 6);
  for (size_t col = 0; col < w; col++) {
    if (8 + 4 * col + 4 > len) corrupt("unfinished image column");
    const size_t off = rd_u32(buf + 8 + 4 * col);
    if (off >= len) corrupt("Invalid image column offset");
    size_t p = off;
    for (;;) {
      if (p >= len) corrupt("unfinished image column");
      const size_t row_start = buf[p++];
      if (row_start == 255) break;
      if (p >= len) corrupt("Missing image run length");
      const size_t run = buf[p++];
      if (row_start + run > h) corrupt("Image run too big");
      if (p >= len) corrupt("Image missing padding byte 1");
      p++;
      if (len - p < run) corrupt("Image source underrun");
      for (size_t k = 0; k < run; k++) img.px_[(row_start + k) * w + col] = buf[p + k];
      p += run;
      if (p >= len) corrupt("Image missing padding byte 2");
      p++;
    }
  }
  return img;
}

void Image::blit(const Image &src, long ox, long oy, bool ignore_transparency) {
  if (ox >= (long)w_ || oy >= (long)h_) return;  // "Fully out of bounds blit" (image.rs:174-180)
  const long y_start = oy < 0 ? -oy : 0;
  const long x_start = ox < 0 ? -ox : 0;
  const long y_end = (long)h_ > (long)src.h_ + oy ? (long)src.h_ : (long)h_ - oy;
  const long x_end = (long)w_ > (long)src.w_ + ox ? (long)src.w_ : (long)w_ - ox;
  // The reference computes x_end - x_start in usize and would overflow when the source lies fully to
  // the left/top; defined here as "nothing to copy".
  if (x_end <= x_start || y_end <= y_start) return;
  for (long y = y_start; y < y_end; y++) {
    const uint16_t *s = &src.px_[(size_t)y * src.w_ + (size_t)x_start];
    uint16_t *d = &px_[(size_t)(y + oy) * w_ + (size_t)(x_start + ox)];
    const size_t n = (size_t)(x_end - x_start);
    if (ignore_transparency) {
      std::memcpy(d, s, n * sizeof(uint16_t));
    } else {
      for (size_t k = 0; k < n; k++) {  // copy where bit 15 of the source is clear (image.rs:243-249)
        const uint16_t blend = (uint16_t)(0u - (uint16_t)(s[k] >> 15));
        d[k] = (uint16_t)((s[k] & (uint16_t)~blend) | (d[k] & blend));
      }
    }
  }
}

// ---- TextureDirectory ------------------------------------------------------------------------------
void TextureDirectory::read_patches(const Archive &wad) {  // tex.rs:358-410
  const size_t li = wad.required_named_lump("PNAMES");
  const uint8_t *buf = wad.lump_data(li);
  const size_t len = wad.lump(li).size;
  if (len < 4) corrupt("Missing number of patches in PNAMES");
  const size_t n = rd_u32(buf);
  for (size_t i = 0; i < n; i++) {
    if (4 + 8 * i + 8 > len) continue;  // "Failed to read patch name" -> skipped
    WadName name;
    try {
      name = WadName::from_bytes(buf + 4 + 8 * i, 8);
    } catch (const WadError &) {
      continue;
    }
    const auto idx = wad.named_lump(name);
    if (!idx) {
      patches_.emplace_back(name, std::nullopt);
      continue;
    }
    try {
      patches_.emplace_back(name, Image::from_buffer(wad.lump_data(*idx), wad.lump(*idx).size));
    } catch (const WadError &) {
      patches_.emplace_back(name, std::nullopt);  // "Skipping patch"
    }
  }
}

void TextureDirectory::read_textures(const uint8_t *buf, size_t len) {  // tex.rs:499-592
  if (len < 4) corrupt("Missing number of textures.");
  const size_t n = rd_u32(buf);
  if (!(n * 4 < len - 4)) corrupt("Textures lump too small for offsets");
  for (size_t i = 0; i < n; i++) {
    const size_t off = rd_u32(buf + 4 + 4 * i);
    if (off >= len) corrupt("Textures lump too small for offsets");
    if (off + 22 > len) continue;  // header unreadable -> texture skipped
    WadName name;
    try {
      name = WadName::from_bytes(buf + off, 8);
    } catch (const WadError &) {
      continue;
    }
    const size_t w = rd_u16(buf + off + 12), h = rd_u16(buf + off + 14), npatch = rd_u16(buf + off + 20);
    if (w > MAX_IMAGE_SIZE || h > MAX_IMAGE_SIZE) continue;
    Image image(w, h);
    size_t p = off + 22;
    for (size_t k = 0; k < npatch; k++) {
      if (p + 10 > len) continue;
      const long ox = rd_i16(buf + p);
      long oy = rd_i16(buf + p + 2);
      const size_t pi = rd_u16(buf + p + 4);
      p += 10;
      if (oy <= 0) oy = 0;  // tex.rs:560-567
      if (pi < patches_.size() && patches_[pi].second) image.blit(*patches_[pi].second, ox, oy, k == 0);
    }
    textures_.insert(name, std::move(image));
  }
}

TextureDirectory TextureDirectory::from_archive(const Archive &wad) {
  TextureDirectory t;
  {
    const size_t pp = wad.required_named_lump("PLAYPAL"), cm = wad.required_named_lump("COLORMAP");
    const size_t npal = wad.checked_count(pp, 768), ncm = wad.checked_count(cm, 256);
    t.palettes_.assign(wad.lump_data(pp), wad.lump_data(pp) + npal * 768);
    t.colormaps_.assign(wad.lump_data(cm), wad.lump_data(cm) + ncm * 256);
  }
  t.read_patches(wad);
  for (const char *lump_name : {"TEXTURE1", "TEXTURE2"}) {
    const auto li = wad.named_lump(WadName::from_str(lump_name));
    if (!li) continue;
    t.read_textures(wad.lump_data(*li), wad.lump(*li).size);
  }
  {  // read_flats (tex.rs:594-606)
    const size_t start = wad.required_named_lump("F_START"), end = wad.required_named_lump("F_END");
    for (size_t i = start; i < end; i++) {
      const LumpInfo &li = wad.lump(i);
      if (li.size == 0) continue;
      t.flats_.insert(li.name, std::vector<uint8_t>(wad.lump_data(i), wad.lump_data(i) + li.size));
    }
  }
  {  // read_sprites (tex.rs:475-497): sprites share the `textures` map
    const size_t start = wad.required_named_lump("S_START") + 1, end = wad.required_named_lump("S_END");
    for (size_t i = start; i < end; i++) {
      try {
        t.textures_.insert(wad.lump(i).name, Image::from_buffer(wad.lump_data(i), wad.lump(i).size));
      } catch (const WadError &) {
        continue;
      }
    }
  }
  t.animated_walls_ = wad.metadata().animated_walls;
  t.animated_flats_ = wad.metadata().animated_flats;
  return t;
}

std::vector<uint8_t> TextureDirectory::build_palette_texture(size_t palette, size_t cm_start, size_t cm_end) const {
  const size_t n = cm_end - cm_start;
  std::vector<uint8_t> mapped(256 * n * 3, 0);
  const uint8_t *pal = this->palette(palette);
  for (size_t i = cm_start; i < cm_end && i < num_colormaps(); i++) {
    const size_t offset = i * 256 * 3;  // absolute colormap index (tex.rs:153)
    for (size_t c = 0; c < 256; c++) {
      if (offset + c * 3 + 3 > mapped.size()) break;
      std::memcpy(&mapped[offset + c * 3], pal + 3 * colormap(i)[c], 3);
    }
  }
  return mapped;
}

namespace {
template <class ImageT>
struct AtlasEntry {
  WadName name;
  const ImageT *image;
  size_t frame_offset, num_frames;
};

// ordered_atlas_entries + search_for_frame (tex.rs:421-473)
template <class ImageT, class Lookup>
std::vector<AtlasEntry<ImageT>> ordered_atlas_entries(const std::vector<std::vector<WadName>> &animations,
                                                      Lookup lookup, const std::vector<WadName> &names) {
  NameIndexMap<const std::vector<WadName> *> by_first;
  for (const WadName &name : names) {
    const std::vector<WadName> *frames = nullptr;
    for (const auto &anim : animations)
      if (std::find(anim.begin(), anim.end(), name) != anim.end()) {
        frames = &anim;
        break;
      }
    by_first.insert(frames ? (*frames)[0] : name, frames);
  }
  std::vector<AtlasEntry<ImageT>> entries;
  for (const auto &kv : by_first.items()) {
    if (kv.second) {
      for (size_t off = 0; off < kv.second->size(); off++)
        if (const ImageT *img = lookup((*kv.second)[off]))
          entries.push_back({(*kv.second)[off], img, off, kv.second->size()});
    } else if (const ImageT *img = lookup(kv.first)) {
      entries.push_back({kv.first, img, 0, 1});
    }
  }
  return entries;
}
}  // namespace

std::pair<TransparentImage, BoundsLookup> TextureDirectory::build_texture_atlas(const std::vector<WadName> &names) const {
  auto entries = ordered_atlas_entries<Image>(animated_walls_, [&](const WadName &n) { return texture(n); }, names);
  std::pair<TransparentImage, BoundsLookup> out;
  if (entries.empty()) return out;
  size_t max_w = 0, num_pixels = 0;
  for (auto &e : entries) {
    max_w = std::max(max_w, e.image->width());
    num_pixels += e.image->width() * e.image->height();
  }
  size_t size[2] = {std::min<size_t>(128, next_pow2(max_w)), 128};
  auto next_size = [&]() {  // tex.rs:186-200
    for (;;) {
      if (size[0] <= size[1]) {
        if (size[0] == 4096) throw WadError(RDOOM_BAD_LEVEL, "Could not fit wall atlas.");
        size[0] *= 2;
        size[1] = 128;
      } else {
        size[1] *= 2;
      }
      if (size[0] * size[1] >= num_pixels) break;
    }
  };
  next_size();
  struct Pos {
    long x, y;
    size_t row_height;
  };
  std::vector<Pos> positions;
  bool transposed = false;
  for (;;) {
    positions.clear();
    size_t ox = 0, oy = 0, row_height = 0;
    bool failed = false;
    for (auto &e : entries) {
      const size_t w = e.image->width(), h = e.image->height();
      if (ox + w > size[0]) {
        ox = 0;
        oy += row_height;
        row_height = 0;
      }
      if (h > row_height) row_height = h;
      if (oy + h > size[1]) {
        failed = true;
        break;
      }
      positions.push_back({(long)ox, (long)oy, row_height});
      ox += w;
    }
    if (!failed) break;
    std::swap(size[0], size[1]);
    transposed = !transposed;
    if (transposed && size[0] != size[1]) continue;
    transposed = false;
    next_size();
  }
  Image atlas(size[0], size[1]);
  for (size_t i = 0; i < entries.size(); i++) {
    atlas.blit(*entries[i].image, positions[i].x, positions[i].y, true);
    // every frame reports frame 0's place (tex.rs:258-261).  An animation whose earlier frames are missing from the
    // WAD has frame_offset > i: the reference panics on the index (usize underflow); here it is a corrupt-WAD error
    if (entries[i].frame_offset > i)
      throw WadError(RDOOM_CORRUPT_WAD, "animated texture is missing its first frames (frame " +
                                            std::to_string(entries[i].frame_offset) + " without its predecessors)");
    const Pos &p = positions[i - entries[i].frame_offset];
    out.second.insert(entries[i].name, Bounds{{(float)p.x, (float)p.y},
                                              {(float)entries[i].image->width(), (float)entries[i].image->height()},
                                              entries[i].num_frames,
                                              p.row_height});
  }
  out.first.w = size[0];
  out.first.h = size[1];
  out.first.pixels = std::move(atlas.pixels());
  return out;
}

std::pair<OpaqueImage, BoundsLookup> TextureDirectory::build_flat_atlas(const std::vector<WadName> &names) const {
  using Flat = std::vector<uint8_t>;
  auto entries = ordered_atlas_entries<Flat>(animated_flats_, [&](const WadName &n) { return flat(n); }, names);
  const size_t n = entries.size();
  const size_t width = next_pow2((size_t)std::ceil(std::sqrt((double)n)) * 64);
  const size_t per_row = width / 64;
  const size_t rows = per_row ? (size_t)std::ceil((double)n / (double)per_row) : 0;
  const size_t height = next_pow2(rows * 64);
  std::pair<OpaqueImage, BoundsLookup> out;
  out.first.w = width;
  out.first.h = height;
  out.first.pixels.assign(width * height, 255);
  size_t row = 0, col = 0;
  float anim_start[2] = {0.0f, 0.0f};
  for (auto &e : entries) {
    const size_t ox = col * 64, oy = row * 64;
    if (e.frame_offset == 0) {
      anim_start[0] = (float)ox;
      anim_start[1] = (float)oy;
    }
    out.second.insert(e.name, Bounds{{anim_start[0], anim_start[1]}, {64.0f, 64.0f}, e.num_frames, 64});
    if (e.image->size() < 4096) throw WadError(RDOOM_CORRUPT_WAD, "flat lump shorter than 4096 bytes");
    for (size_t y = 0; y < 64; y++)
      std::memcpy(&out.first.pixels[ox + (y + oy) * width], e.image->data() + y * 64, 64);
    if (++col == per_row) {
      col = 0;
      row++;
    }
  }
  return out;
}

}  // namespace rdoom::wad
