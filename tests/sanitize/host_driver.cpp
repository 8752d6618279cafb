// Sanitizer driver of the HOST half (csrc/host/*.cpp) -- test infrastructure.
// Built by tests/test_host_sanitizers.py with g++ -fsanitize=address,undefined over the product's own host sources; no HIP
// (the device tessellation hooks are never requested: use_gpu_tessellation = 0, the CPU-only path of BASELINE config 1).
//   host_driver <iwad> <metadata> <first level> <last level>
// prints one line per step: "OPEN <status>", then per level "LEVEL <i> <status> <counters...> <crc32 over every array>".
// A sanitizer report aborts the process (non-zero exit, report on stderr); malformed input must instead end in a status
// code, or -- where the reference warns and skips (wad/src/visitor.rs:599-604, 622-643, 718-729, 855-872) -- in a level.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "game_level.hpp"
#include "rdoom.h"

namespace rdoom::game {
std::vector<std::vector<wad::Pnt2f>> tessellate_on_device(const wad::Level &, const std::vector<wad::LevelWalker::LeafInput> &) {
  std::abort();  // never requested here
}
std::vector<wad::SegGeometry> tessellate_segs_on_device(const std::vector<wad::SegInput> &) { std::abort(); }
}  // namespace rdoom::game

// zlib's crc32 (reflected 0xEDB88320), so that the Python side can run the same digest over the oracle's arrays
static uint32_t crc(uint32_t c, const void *p, size_t n) {
  static uint32_t table[256];
  if (!table[1])
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t k = i;
      for (int j = 0; j < 8; j++) k = (k >> 1) ^ (0xEDB88320u & (0u - (k & 1u)));
      table[i] = k;
    }
  const unsigned char *b = static_cast<const unsigned char *>(p);
  c = ~c;
  for (size_t i = 0; i < n; i++) c = table[(c ^ b[i]) & 0xFFu] ^ (c >> 8);
  return ~c;
}

int main(int argc, char **argv) {
  if (argc < 5) return 2;
  rdoom_wad *wad = nullptr;
  const rdoom_status so = rdoom_wad_open(argv[1], argv[2], &wad);
  std::printf("OPEN %d\n", (int)so);
  if (so != RDOOM_OK) return 0;
  uint32_t n_levels = 0;
  rdoom_wad_num_levels(wad, &n_levels);
  std::printf("LEVELS %u\n", n_levels);
  for (int i = std::atoi(argv[3]); i <= std::atoi(argv[4]); i++) {
    rdoom_built *built = nullptr;
    const rdoom_status sb = rdoom_wad_build_level(wad, (uint32_t)i, 0, &built);
    if (sb != RDOOM_OK) {
      std::printf("LEVEL %d %d\n", i, (int)sb);
      continue;
    }
    rdoom_level_desc d;
    rdoom_counters c;
    rdoom_built_desc(built, &d);
    rdoom_built_counters(built, &c);
    uint32_t h = 0;
    h = crc(h, d.static_verts, (size_t)d.n_static_verts * sizeof(rdoom_static_vertex));
    h = crc(h, d.static_indices, (size_t)d.n_static_indices * 4);
    h = crc(h, d.sky_verts, (size_t)d.n_sky_verts * 12);
    h = crc(h, d.sky_indices, (size_t)d.n_sky_indices * 4);
    h = crc(h, d.decor_verts, (size_t)d.n_decor_verts * sizeof(rdoom_sprite_vertex));
    h = crc(h, d.decor_indices, (size_t)d.n_decor_indices * 4);
    h = crc(h, d.draws, (size_t)d.n_draws * sizeof(rdoom_draw));
    h = crc(h, d.flat_atlas, (size_t)d.flat_w * d.flat_h);
    h = crc(h, d.wall_atlas, (size_t)d.wall_w * d.wall_h * 2);
    h = crc(h, d.decor_atlas, (size_t)d.decor_w * d.decor_h * 2);
    h = crc(h, d.sky_texture, (size_t)d.sky_w * d.sky_h * 2);
    h = crc(h, d.colormap, 32 * 256);
    uint8_t lights[256];
    rdoom_built_lights_at(built, 1.25f, lights);
    h = crc(h, lights, 256);
    std::printf("LEVEL %d 0 verts=%u idx=%u sky=%u decor=%u draws=%u %08x\n", i, d.n_static_verts, d.n_static_indices,
                d.n_sky_verts, d.n_decor_verts, d.n_draws, h);
    rdoom_built_destroy(built);
  }
  rdoom_wad_close(wad);
  return 0;
}
