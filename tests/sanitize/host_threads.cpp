// ThreadSanitizer driver of the HOST half (csrc/host/*.cpp) -- test infrastructure (tests/test_host_threads.py builds it with
// g++ -fsanitize=thread over the product's own host sources; no HIP).  The threading contract of SURVEY 8(b) / INTEGRATION.md:
// a rdoom_wad handle is single-owner (the reference's Archive is !Sync, wad/src/archive.rs:21), nothing is shared between
// handles, rdoom_last_error is thread-local, rdoom_debug_set may be called while other threads work.
//   host_threads <iwad> <metadata> <threads> <rounds>
// Every thread opens its OWN handle, builds every level `rounds` times and digests the arrays; one of them provokes errors
// (a level index out of range) in between and checks that its message is its own; the main thread toggles a debug hook
// meanwhile.  Prints "DIGEST <thread> <crc>" per thread: the digests must all be equal (the test checks), TSan must be silent.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "game_level.hpp"
#include "rdoom.h"

namespace rdoom::game {
std::vector<std::vector<wad::Pnt2f>> tessellate_on_device(const wad::Level &, const std::vector<wad::LevelWalker::LeafInput> &) {
  std::abort();  // never requested here
}
std::vector<wad::SegGeometry> tessellate_segs_on_device(const std::vector<wad::SegInput> &) { std::abort(); }
}  // namespace rdoom::game

static uint32_t crc(uint32_t c, const void *p, size_t n) {
  const unsigned char *b = static_cast<const unsigned char *>(p);
  c = ~c;
  for (size_t i = 0; i < n; i++) {
    c ^= b[i];
    for (int j = 0; j < 8; j++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
  }
  return ~c;
}

int main(int argc, char **argv) {
  if (argc < 5) return 2;
  const int threads = std::atoi(argv[3]), rounds = std::atoi(argv[4]);
  std::vector<uint32_t> digest((size_t)threads, 0u);
  std::atomic<int> failures{0}, running{threads};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; t++)
    pool.emplace_back([&, t] {
      rdoom_wad *wad = nullptr;
      if (rdoom_wad_open(argv[1], argv[2], &wad) != RDOOM_OK) {
        failures++;
        running--;
        return;
      }
      uint32_t n_levels = 0, h = 0;
      rdoom_wad_num_levels(wad, &n_levels);
      for (int r = 0; r < rounds; r++)
        for (uint32_t i = 0; i < n_levels; i++) {
          if (t == 0) {  // an error on THIS thread: the message must be this thread's own when it is read back
            rdoom_built *none = nullptr;
            if (rdoom_wad_build_level(wad, n_levels + 7u + (uint32_t)r, 0, &none) == RDOOM_OK) failures++;
            if (!std::strstr(rdoom_last_error(), "level index out of range")) failures++;
          }
          rdoom_built *built = nullptr;
          if (rdoom_wad_build_level(wad, i, 0, &built) != RDOOM_OK) {
            failures++;
            continue;
          }
          rdoom_level_desc d;
          rdoom_built_desc(built, &d);
          h = crc(h, d.static_verts, (size_t)d.n_static_verts * sizeof(rdoom_static_vertex));
          h = crc(h, d.static_indices, (size_t)d.n_static_indices * 4);
          h = crc(h, d.wall_atlas, (size_t)d.wall_w * d.wall_h * 2);
          h = crc(h, d.flat_atlas, (size_t)d.flat_w * d.flat_h);
          uint8_t lights[256];
          rdoom_built_lights_at(built, 0.5f + (float)r, lights);
          h = crc(h, lights, 256);
          rdoom_built_destroy(built);
          if (t != 0 && std::strstr(rdoom_last_error(), "level index")) failures++;  // thread 0's message leaked in
        }
      digest[(size_t)t] = h;
      rdoom_wad_close(wad);
      running--;
    });
  int flips = 0;
  while (running.load() > 0) {  // the debug hooks are a locked snapshot: setting one while others build must not race
    rdoom_debug_set("frag_chunk", flips++ & 7);
    std::this_thread::yield();
  }
  rdoom_debug_set("reset", 0);
  for (auto &th : pool) th.join();
  for (int t = 0; t < threads; t++) std::printf("DIGEST %d %08x\n", t, digest[(size_t)t]);
  std::printf("FAILURES %d FLIPS %d\n", failures.load(), flips);
  return failures.load() ? 1 : 0;
}
