// C ABI of the loader + builder (include/rdoom.h "loader + builder").  TEMPORARY: not implemented yet.
#include "../common.hpp"

extern "C" {
#define NOT_YET return rdoom::fail(RDOOM_BAD_ARG, "%s: not implemented yet", __func__)
rdoom_status rdoom_wad_open(const char *, const char *, rdoom_wad **) { NOT_YET; }
void rdoom_wad_close(rdoom_wad *) {}
rdoom_status rdoom_wad_num_levels(const rdoom_wad *, uint32_t *) { NOT_YET; }
rdoom_status rdoom_wad_level_name(const rdoom_wad *, uint32_t, char *) { NOT_YET; }
rdoom_status rdoom_wad_name_from_bytes(const uint8_t *, uint32_t, uint8_t *) { NOT_YET; }
rdoom_status rdoom_wad_build_level(const rdoom_wad *, uint32_t, int32_t, rdoom_built **) { NOT_YET; }
void rdoom_built_destroy(rdoom_built *) {}
rdoom_status rdoom_built_desc(const rdoom_built *, rdoom_level_desc *) { NOT_YET; }
rdoom_status rdoom_built_counters(const rdoom_built *, rdoom_counters *) { NOT_YET; }
rdoom_status rdoom_built_lights_at(const rdoom_built *, float, uint8_t *) { NOT_YET; }
rdoom_status rdoom_built_start(const rdoom_built *, float *, float *) { NOT_YET; }
rdoom_status rdoom_built_floor_centroids(const rdoom_built *, const float **, uint32_t *) { NOT_YET; }
}
