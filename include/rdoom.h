/* rdoom.h -- C ABI of the MI355X pose-batch renderer for Doom WAD levels.
 *
 * This is the drop-in boundary.  It replaces what rust-doom hands to
 * glium/OpenGL; every entry point cites the reference interface it stands in for.  Plain pointers
 * and sizes only; no C++/torch types; nothing throws across it.
 *
 *   loader + builder  (host C++, mirrors `wad` + `game::level`):   rdoom_wad_*, rdoom_built_*
 *   device renderer   (hand-written HIP, gfx950):                   rdoom_level_*, rdoom_batch_*
 *
 * Conventions
 *   - every function returns rdoom_status (0 = ok, <0 = error); rdoom_last_error() gives a
 *     thread-local message (reference: Result<T, failchain::BoxedError<ErrorKind>>,
 *     wad/src/errors.rs:6-19; visitor callbacks are infallible, bad level data is skipped).
 *   - matrices are column-major float[16], exactly the GLSL uniforms u_modelview / u_projection
 *     (engine/src/uniforms.rs:273-280).
 *   - framebuffers are 8-bit palette indices, row 0 = bottom row (glReadPixels order), background 0.
 *   - a rdoom_level is immutable after create (shareable); a rdoom_batch is single-owner.
 */
#ifndef RDOOM_H
#define RDOOM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t rdoom_status;
#define RDOOM_OK 0
#define RDOOM_BAD_ARG (-1)
#define RDOOM_HIP_ERROR (-2)
#define RDOOM_OOM (-3)
#define RDOOM_BAD_LEVEL (-4)
#define RDOOM_IO (-5)        /* ErrorKind::Io            (wad/src/errors.rs:9-19) */
#define RDOOM_CORRUPT_WAD (-6)  /* ErrorKind::CorruptWad */
#define RDOOM_CORRUPT_META (-7) /* ErrorKind::CorruptMetadata */

/* draw kinds, in the reference's per-object attach order (game/src/level.rs:443-496) */
#define RDOOM_KIND_FLAT 0u
#define RDOOM_KIND_WALL 1u
#define RDOOM_KIND_DECOR 2u
#define RDOOM_KIND_SKY 3u
#define RDOOM_ALL_KINDS 0xFu

/* game/src/vertex.rs:5-16  StaticVertex (repr(C), 48 bytes) */
typedef struct rdoom_static_vertex {
  float a_pos[3];
  float a_atlas_uv[2];
  float a_tile_uv[2];
  float a_tile_size[2];
  float a_scroll_rate;
  float a_row_height;
  uint8_t a_num_frames;
  uint8_t a_light;
  uint8_t _pad[2];
} rdoom_static_vertex;

/* game/src/vertex.rs:30-40  SpriteVertex (repr(C), 44 bytes) */
typedef struct rdoom_sprite_vertex {
  float a_pos[3];
  float a_atlas_uv[2];
  float a_tile_uv[2];
  float a_tile_size[2];
  float a_local_x;
  uint8_t a_num_frames;
  uint8_t a_light;
  uint8_t _pad[2];
} rdoom_sprite_vertex;

/* One `frame.draw(mesh, indices, program, ...)` of the reference (engine/src/renderer.rs:100-157):
 * a range of the index array of its vertex buffer.  Draw order == array order. */
typedef struct rdoom_draw {
  uint32_t kind;        /* RDOOM_KIND_* */
  uint32_t object_id;   /* wad::ObjectId (visitor.rs:142-143); 0 = static world */
  uint32_t first_index; /* into static_indices / decor_indices / sky_indices by kind */
  uint32_t index_count; /* multiple of 3 (TrianglesList, engine/src/meshes.rs:97-106) */
} rdoom_draw;

/* Everything Builder::build + GameShaders::load_level give glium for one level
 * (game/src/level.rs:424-496, game/src/game_shaders.rs:175-453).  Caller owns all arrays;
 * rdoom_level_create copies them to the device. */
typedef struct rdoom_level_desc {
  const rdoom_static_vertex *static_verts;
  uint32_t n_static_verts;
  const uint32_t *static_indices;
  uint32_t n_static_indices;
  const float *sky_verts; /* SkyVertex: xyz triples (vertex.rs:53-57) */
  uint32_t n_sky_verts;
  const uint32_t *sky_indices;
  uint32_t n_sky_indices;
  const rdoom_sprite_vertex *decor_verts;
  uint32_t n_decor_verts;
  const uint32_t *decor_indices;
  uint32_t n_decor_indices;
  const rdoom_draw *draws;
  uint32_t n_draws;
  const uint8_t *flat_atlas; /* wad::OpaqueImage (tex.rs:47-50), U8, REPEAT/NEAREST */
  uint32_t flat_w, flat_h;   /* powers of two (tex.rs:281-286) */
  const uint16_t *wall_atlas; /* wad::TransparentImage (tex.rs:42-45), U8U8: lo=index, hi>=0x80 transparent */
  uint32_t wall_w, wall_h;    /* powers of two (tex.rs:183-200) */
  const uint16_t *decor_atlas;
  uint32_t decor_w, decor_h;
  const uint16_t *sky_texture; /* game_shaders.rs:358-387 */
  uint32_t sky_w, sky_h;
  float sky_tiled_band_size;
  const uint8_t *playpal;  /* 768 bytes, palette 0 (kept for RGB expansion by callers) */
  const uint8_t *colormap; /* 32*256 bytes: rows of build_palette_texture(0,0,32) before the PLAYPAL map (tex.rs:137-166) */
} rdoom_level_desc;

/* Per-frame uniforms (engine/src/renderer.rs:78-132, game/src/game_shaders.rs:84-92). */
typedef struct rdoom_pose {
  float modelview[16];
  float projection[16];
  float time; /* u_time, seconds */
  float _pad;
} rdoom_pose;

typedef struct rdoom_level rdoom_level;
typedef struct rdoom_batch rdoom_batch;
typedef struct rdoom_wad rdoom_wad;
typedef struct rdoom_built rdoom_built;

/* per-kernel GPU times of the last rdoom_batch_render, from hipEvents on the render stream */
typedef struct rdoom_timings {
  float setup_ms, raster_ms, fragment_ms, total_ms;
  uint64_t pixels; /* n_poses * width * height of that render */
  uint64_t visible_triangles;
  uint64_t fixup_pixels; /* pixels re-resolved by the alpha-leak fixup kernel (normally a handful) */
} rdoom_timings;

/* Host-side times of the load and build phases, the ones the reference logs with one-shot Instants (wad/src/tex.rs:67-88,
 * 371-408, 479-495; game/src/level.rs:333, 384-396).  Single thread, wall clock (std::chrono::steady_clock).
 * rdoom_wad_timings fills open_ms + textures_ms (the others are 0); rdoom_built_timings fills the other four. */
typedef struct rdoom_host_timings {
  float open_ms;        /* Archive::open: header, directory, metadata (archive.rs:36-106, meta.rs:143-154) */
  float textures_ms;    /* TextureDirectory::from_archive: PLAYPAL, COLORMAP, patches, TEXTURE1/2, flats, sprites (tex.rs:53-107) */
  float level_lumps_ms; /* Level::from_archive: the eight lumps of the level (level.rs:34-81) */
  float atlases_ms;     /* build_flat_atlas + build_texture_atlas (walls, decor) + sky (game_shaders.rs:175-387) */
  float analysis_ms;    /* LevelAnalysis::new (visitor.rs:323-444) */
  float walk_ms;        /* LevelWalker::walk driving the Builder, then the index lists (level.rs:330-496) */
} rdoom_host_timings;

/* counters logged by the reference at level build (game/src/level.rs:384-422) */
typedef struct rdoom_counters {
  uint32_t num_wall_quads, num_floor_polys, num_ceil_polys, num_sky_wall_quads, num_sky_floor_polys,
      num_sky_ceil_polys, num_decors, num_static_tris, num_sky_tris, num_sprite_tris, num_objects, num_lights;
} rdoom_counters;

/* Which paths the last render of a batch took (read back from the device after it): a workload whose poses overflow their
 * tile lists is rasterised from the sorted list -- correct, and slow -- and nothing else would say so.  No reference counterpart. */
typedef struct rdoom_path_stats {
  uint32_t poses;                 /* of the last render */
  uint32_t bins_overflowed_poses; /* poses whose tile lists did not fit (or a frame with too many tiles): rasterised from the sorted list */
  uint64_t tiles;                 /* 64 x 64 tiles of those poses' frames */
  uint64_t split_tiles;           /* tiles whose list (more than 64 entries) is stored per 32 x 32 quadrant */
  uint64_t tile_entries;          /* sum of the tile lists' lengths (one per triangle and tile) */
  uint64_t quadrants;             /* 32 x 32 quadrants that lie (partly) inside the frame */
  uint64_t described_quadrants;   /* of those: all pixels show one record -- no visibility words stored or read */
} rdoom_path_stats;

const char *rdoom_last_error(void);

/* ---- devices ------------------------------------------------------------------------------ */
rdoom_status rdoom_device_count(int32_t *out_count);
rdoom_status rdoom_set_device(int32_t device);

/* ---- device renderer: replaces engine Meshes/Uniforms uploads + Renderer::update ----------- */
/* replaces VertexBuffer::immutable / IndexBuffer::persistent / Texture2d::new uploads
 * (engine/src/meshes.rs:126-201, engine/src/uniforms.rs:146-221) */
rdoom_status rdoom_level_create(const rdoom_level_desc *desc, rdoom_level **out_level);
void rdoom_level_destroy(rdoom_level *level);
/* Several levels resident together, as ONE rdoom_level handle (rdoom_level_create = a set of one): a pose batch may then mix
 * poses of different levels in one render (rdoom_batch_render_levels) -- one launch set for, say, the 9 x 128 poses one of eight
 * GPUs renders of E1M1..E1M9 instead of nine small ones.  The reference keeps one level loaded at a time and draws it frame by
 * frame (game/src/level.rs:330-496 builds and uploads it, engine/src/renderer.rs:98-157 is the draw loop); this replaces N of
 * those uploads.  The levels must come from one IWAD: their COLORMAP tables must be identical (wad/src/tex.rs:137-166 reads the
 * archive's one COLORMAP lump).  Primitive ids, object ids and light tables stay per level.  At most 2^26 atlas texels in the set. */
rdoom_status rdoom_levelset_create(const rdoom_level_desc *const *descs, uint32_t n_levels, rdoom_level **out_level);
/* how many levels the handle holds (1 for rdoom_level_create) */
rdoom_status rdoom_level_num_levels(const rdoom_level *level, uint32_t *out);

/* allocates the device scratch for up to max_poses frames of width x height -- any size up to 16384 on a side, as the
 * reference's --resolution WxH (src/main.rs:41).  Rows of the device framebuffer are rdoom_batch_framebuffer_pitch bytes
 * apart: the width itself when it is a multiple of 4, else the next multiple of 8. */
rdoom_status rdoom_batch_create(const rdoom_level *level, uint32_t width, uint32_t height, uint32_t max_poses,
                                rdoom_batch **out_batch);
void rdoom_batch_destroy(rdoom_batch *batch);

/* replaces the frame.draw loop of Renderer::update (engine/src/renderer.rs:98-157) for n_poses
 * frames at once.  lights: n_poses tables of 256 bytes (Lights::fill_buffer_at, game/src/lights.rs:26-30)
 * spaced lights_stride bytes apart (0 = one shared table).  kinds_mask selects draw kinds.
 * Asynchronous on `stream` (a hipStream_t, may be NULL); results stay on the device.  A batch's scratch is written by every
 * render: consecutive renders of ONE batch must be ordered -- the same stream, or streams the caller orders -- and the host may
 * queue two of them before it waits for the first one's pose copy (pinned staging, two deep). */
rdoom_status rdoom_batch_render(rdoom_batch *batch, const rdoom_pose *poses, const uint8_t *lights,
                                uint32_t lights_stride, uint32_t n_poses, uint32_t kinds_mask, void *stream);
/* same, with per-kernel hipEvent timing (synchronises the stream) */
rdoom_status rdoom_batch_render_timed(rdoom_batch *batch, const rdoom_pose *poses, const uint8_t *lights,
                                      uint32_t lights_stride, uint32_t n_poses, uint32_t kinds_mask, void *stream,
                                      rdoom_timings *out);
/* Asynchronous like rdoom_batch_render, but the four hipEvents around the kernels are kept (up to 64 renders may be
 * pending); rdoom_batch_collect_timings waits for the last of them and returns the SUMS over the pending renders
 * (pixels = all their pixels; visible_triangles / fixup_pixels of the last one) and their number.  For profiling a
 * pipelined sequence of renders without a host synchronisation in between (bench.py). */
rdoom_status rdoom_batch_render_profiled(rdoom_batch *batch, const rdoom_pose *poses, const uint8_t *lights,
                                         uint32_t lights_stride, uint32_t n_poses, uint32_t kinds_mask, void *stream);
rdoom_status rdoom_batch_collect_timings(rdoom_batch *batch, rdoom_timings *out_sums, uint32_t *out_renders);
/* Same with moving objects (doors, lifts): the reference sets u_modelview = view o model transform for the draws
 * of each object (engine/src/renderer.rs:120-132; game/src/level.rs:203-255 moves the transforms).
 * object_modelviews: n_poses x n_objects column-major matrices, entry [p][o] = the u_modelview of object o's draws
 * in frame p (object 0 = the static world; an object at rest has the pose's own modelview).
 * n_objects >= rdoom_level_num_objects. */
rdoom_status rdoom_batch_render_objects(rdoom_batch *batch, const rdoom_pose *poses, const uint8_t *lights,
                                        uint32_t lights_stride, uint32_t n_poses, uint32_t kinds_mask, void *stream,
                                        const float *object_modelviews, uint32_t n_objects);
/* rdoom_batch_render / _objects / _profiled for a batch created on a level SET: pose p is a view of level level_of_pose[p]
 * (an index into the descs of rdoom_levelset_create), lights + p * lights_stride is THAT level's light table at the pose's time.
 * object_modelviews may be NULL (no moving objects); else n_poses x n_objects matrices as for rdoom_batch_render_objects, with
 * n_objects >= rdoom_level_num_objects of the set (entries of objects the pose's level does not draw are ignored).
 * flags: RDOOM_RENDER_PROFILED keeps the per-kernel events pending as rdoom_batch_render_profiled does.
 * Replaces the frame.draw loops of Renderer::update (engine/src/renderer.rs:98-157) of SEVERAL loaded levels at once. */
#define RDOOM_RENDER_PROFILED 1u
rdoom_status rdoom_batch_render_levels(rdoom_batch *batch, const rdoom_pose *poses, const uint32_t *level_of_pose, const uint8_t *lights,
                                       uint32_t lights_stride, uint32_t n_poses, uint32_t kinds_mask, void *stream,
                                       const float *object_modelviews, uint32_t n_objects, uint32_t flags);
/* 1 + the largest rdoom_draw.object_id of the level (of a set: of any of its levels) */
rdoom_status rdoom_level_num_objects(const rdoom_level *level, uint32_t *out);
/* Waits for the batch's last render -- on the stream it was queued on; work of other batches on other streams is not waited
 * for -- and returns ITS status: the asynchronous rdoom_batch_render cannot report what only
 * the device finds out (today: the alpha-leak fixup list overflowing, which would leave leaked transparent texels in
 * the frames).  Consumers of rdoom_batch_framebuffer_device call this -- or any of the rdoom_batch_read_* -- before
 * trusting the frames; glFinish is the nearest reference counterpart. */
rdoom_status rdoom_batch_finish(rdoom_batch *batch);

/* waits for the batch's last render like rdoom_batch_finish, then counts (see rdoom_path_stats) */
rdoom_status rdoom_batch_path_stats(rdoom_batch *batch, rdoom_path_stats *out);

/* device pointer to the n_poses palette-index framebuffers of the last render: frame i starts at byte i * height * pitch,
 * row y of it (row 0 = the bottom row, as glReadPixels) at y * pitch, `width` bytes of pixels, then padding */
rdoom_status rdoom_batch_framebuffer_device(const rdoom_batch *batch, uint8_t **out_device_ptr);
/* bytes between consecutive rows of the device framebuffer (== width when width % 4 == 0) */
rdoom_status rdoom_batch_framebuffer_pitch(const rdoom_batch *batch, uint32_t *out_pitch);
/* glReadPixels analogue: waits for the batch's last render (on its stream -- not for the device), copies frames
 * [first, first+count) to host memory, tightly packed (width bytes per row) */
rdoom_status rdoom_batch_read_framebuffer(rdoom_batch *batch, uint32_t first, uint32_t count, uint8_t *host_out);
/* Debug / test facility: capture the winning primitive id per pixel (triangle index in the draw order of the pose's level,
 * 0xFFFFFFFF = none) on the following renders, then read it back.  No GL counterpart. */
rdoom_status rdoom_batch_enable_primitive_ids(rdoom_batch *batch);
rdoom_status rdoom_batch_read_primitive_ids(rdoom_batch *batch, uint32_t first, uint32_t count, uint32_t *host_out);

/* On-device verification of the exact short forms the fragment kernel uses instead of IEEE division
 * (rust-doom_amd/csrc/hip/fastmath.hpp).  No reference counterpart: it certifies that the kernels evaluate
 * static.frag:18-28's divisions and mod() to the same bits as the plain operations.
 * out_counts: [0] 1/x mismatches, [1] 0.9/x mismatches, [2] inputs swept, [3] mod-certificate violations,
 * [4] mod samples, [5] samples certified, [6] samples where the short form's floor differs (all rejected),
 * [7] packed-vs-scalar mismatches.  [0], [1], [3], [7] must be 0. */
rdoom_status rdoom_selftest_fastmath(uint64_t out_counts[8]);

/* Test hooks (no reference counterpart; the library never reads the environment).  Each option selects a differently
 * shaped but EQUIVALENT path through the kernels -- the image must not change -- so that the rarely taken ones can be
 * forced (tests/test_gpu_debug_paths.py).  Process-wide; read when a batch is created ("vis32", "entry_cap") or
 * rendered (the rest).  Names: no_bins, entry_cap, vis32, leak_mod, frag_nq, frag_bw, frag_chunk, bin_threads,
 * no_cover, no_pair, no_settle, settle_max, raster_stats, no_qtab, keep_vis, qpath, no_split (rust-doom_amd/csrc/common.hpp:
 * DebugOptions); "reset" restores the defaults. */
rdoom_status rdoom_debug_set(const char *name, int32_t value);

/* ---- loader + builder: the `wad` crate and `game::level` static-geometry builder ----------- */
/* Archive::open (wad/src/archive.rs:36-60) + TextureDirectory::from_archive (wad/src/tex.rs:53-107) */
rdoom_status rdoom_wad_open(const char *wad_path, const char *metadata_path, rdoom_wad **out_wad);
void rdoom_wad_close(rdoom_wad *wad);
rdoom_status rdoom_wad_timings(const rdoom_wad *wad, rdoom_host_timings *out);    /* how long rdoom_wad_open's phases took */
rdoom_status rdoom_wad_num_levels(const rdoom_wad *wad, uint32_t *out);           /* Archive::num_levels */
rdoom_status rdoom_wad_level_name(const rdoom_wad *wad, uint32_t index, char out_name[9]); /* WadSystem::level_name */
/* WadName::from_bytes (wad/src/name.rs:41-75); out = 8 bytes */
rdoom_status rdoom_wad_name_from_bytes(const uint8_t *bytes, uint32_t len, uint8_t out[8]);

/* ---- the reference's own plug-in point: trait wad::LevelVisitor (wad/src/visitor.rs:65-127) -------------------
 * Thirteen callbacks, every one optional (NULL = the trait's default: do nothing).  Payloads mirror the structs of
 * visitor.rs:24-63 and are BORROWED for the duration of the call (the walker reuses its scratch per sub-sector,
 * visitor.rs:646-651): copy out what you keep.  Callbacks are infallible, as in the reference, and must not unwind
 * or longjmp across the library.  `user` is handed back untouched. */
typedef struct rdoom_light_info {   /* wad/src/light.rs:8-25 LightInfo { level, effect: Option<LightEffect> } */
  float level;
  int32_t has_effect;               /* 0: effect is None, the rest is unset */
  int32_t effect_kind;              /* 0 Glow, 1 Random, 2 Alternate (LightEffectKind) */
  float alt_level, speed, duration, sync;
} rdoom_light_info;
typedef struct rdoom_static_quad {  /* visitor.rs:24-34 */
  uint32_t object_id;
  float v1[2], v2[2];
  float tex_start[2], tex_end[2], height_range[2];
  const rdoom_light_info *light_info;
  float scroll;
  int32_t has_tex_name;             /* Option<WadName> */
  uint8_t tex_name[8];
  int32_t blocker;
} rdoom_static_quad;
typedef struct rdoom_static_poly {  /* visitor.rs:36-42 */
  uint32_t object_id;
  const float *vertices;            /* n_vertices x (x, z) */
  uint32_t n_vertices;
  float height;
  const rdoom_light_info *light_info;
  uint8_t tex_name[8];
} rdoom_static_poly;
typedef struct rdoom_sky_quad {     /* visitor.rs:44-48 */
  uint32_t object_id;
  float v1[2], v2[2], height_range[2];
} rdoom_sky_quad;
typedef struct rdoom_sky_poly {     /* visitor.rs:50-54 */
  uint32_t object_id;
  const float *vertices;
  uint32_t n_vertices;
  float height;
} rdoom_sky_poly;
typedef struct rdoom_decor {        /* visitor.rs:56-63 */
  uint32_t object_id;
  float low[3], high[3], half_width;
  const rdoom_light_info *light_info;
  uint8_t tex_name[8];
} rdoom_decor;
typedef struct rdoom_line2f {       /* math/src/line.rs:5-10 Line2 { origin, displace, length } */
  float origin[2], displace[2], length;
} rdoom_line2f;
enum { RDOOM_MARKER_START_POS = 0, RDOOM_MARKER_TELEPORT_START = 1, RDOOM_MARKER_TELEPORT_END = 2 };  /* visitor.rs:129-133 */
enum { RDOOM_BRANCH_POSITIVE = 0, RDOOM_BRANCH_NEGATIVE = 1 };                                        /* visitor.rs:135-139 */
typedef struct rdoom_visitor_vtbl { /* trait LevelVisitor, method for method (visitor.rs:65-116) */
  void (*visit_wall_quad)(void *user, const rdoom_static_quad *quad);
  void (*visit_floor_poly)(void *user, const rdoom_static_poly *poly);
  void (*visit_ceil_poly)(void *user, const rdoom_static_poly *poly);
  void (*visit_floor_sky_poly)(void *user, const rdoom_sky_poly *poly);
  void (*visit_ceil_sky_poly)(void *user, const rdoom_sky_poly *poly);
  void (*visit_sky_quad)(void *user, const rdoom_sky_quad *quad);
  void (*visit_marker)(void *user, const float pos[3], float yaw_rad, int32_t marker, uint32_t player);
  void (*visit_decor)(void *user, const rdoom_decor *decor);
  void (*visit_bsp_root)(void *user, const rdoom_line2f *line);
  void (*visit_bsp_node)(void *user, const rdoom_line2f *line, int32_t branch);
  void (*visit_bsp_leaf)(void *user, int32_t branch);
  void (*visit_bsp_leaf_end)(void *user);
  void (*visit_bsp_node_end)(void *user);
} rdoom_visitor_vtbl;
/* WadSystem::walk<V: LevelVisitor>(&self, &mut V) (game/src/wad_system.rs:47-56): LevelWalker::walk over one level with
 * the caller's visitor only -- the way game::world::WorldBuilder is driven (game/src/world.rs:306). */
rdoom_status rdoom_wad_walk(const rdoom_wad *wad, uint32_t level_index, const rdoom_visitor_vtbl *visitor, void *user);

/* WadSystem::create's level half + GameShaders::load_level + Builder::build
 * (game/src/wad_system.rs:72-113, game/src/game_shaders.rs:175-387, game/src/level.rs:330-496).
 * use_gpu_tessellation != 0 runs the SSECTOR->polygon / SEG->quad kernels on the current device
 * (results are identical to the host walk). */
rdoom_status rdoom_wad_build_level(const rdoom_wad *wad, uint32_t level_index, int32_t use_gpu_tessellation,
                                   rdoom_built **out_built);
/* The same with a second visitor chained AFTER the Builder -- `builder.chain(&mut world_builder)` of
 * game/src/level.rs:378-382 (VisitorChain, visitor.rs:1261-1331): each event reaches the Builder, then `visitor`. */
rdoom_status rdoom_wad_build_level_chained(const rdoom_wad *wad, uint32_t level_index, int32_t use_gpu_tessellation,
                                           const rdoom_visitor_vtbl *visitor, void *user, rdoom_built **out_built);
void rdoom_built_destroy(rdoom_built *built);
/* borrowed pointers into `built`, valid until rdoom_built_destroy */
rdoom_status rdoom_built_desc(const rdoom_built *built, rdoom_level_desc *out_desc);
rdoom_status rdoom_built_counters(const rdoom_built *built, rdoom_counters *out);
rdoom_status rdoom_built_timings(const rdoom_built *built, rdoom_host_timings *out); /* how long the build's phases took */
/* Lights::fill_buffer_at (game/src/lights.rs:26-30) */
rdoom_status rdoom_built_lights_at(const rdoom_built *built, float time, uint8_t out_lights[256]);
/* Builder::visit_marker start pose (game/src/level.rs:757-762) */
rdoom_status rdoom_built_start(const rdoom_built *built, float out_pos[3], float *out_yaw);
/* bounds of the sub-sector floor polygons, for pose generators: n polygons, centroid xz + floor y */
rdoom_status rdoom_built_floor_centroids(const rdoom_built *built, const float **out_xyz, uint32_t *out_n);

/* Camera of the reference: view = inverse(T(eye) * Ry(yaw) * Rx(pitch)),
 * projection = perspective(65 deg, (w/h)*1.2, 0.01, 100) (game/src/player.rs:84-89, 325-345;
 * engine/src/projections.rs:93-101; engine/src/renderer.rs:78-87). */
rdoom_status rdoom_pose_look(const float eye[3], float yaw, float pitch, uint32_t width, uint32_t height, float time,
                             rdoom_pose *out_pose);
/* The same camera in the REFERENCE'S OWN ARITHMETIC, binary32 throughout: the player entity's Decomposed { rot =
 * Quaternion::from(Euler { x: pitch, y: yaw, z: 0 }), disp = pos } (game/src/player.rs:124-131) concatenated with the camera
 * child at (0, 0.12, 0) (player.rs:325-335, engine/src/transforms.rs:121), inverted and turned into a matrix as
 * Renderer::update does (engine/src/renderer.rs:78-87), cgmath 0.18.0's published formulas restated; the projection from
 * f = cot(fovy / 2) in binary32.  `pos` is the PLAYER's position (level.start_pos()), not the eye: the camera height is added
 * here.  rdoom_pose_look (double precision, rounded once) stays the helper of the seeded sweeps; the two agree to a few
 * units in the last place (tests/test_pose_helpers.py). */
rdoom_status rdoom_pose_from_player(const float pos[3], float yaw, float pitch, uint32_t width, uint32_t height, float time,
                                    rdoom_pose *out);

#ifdef __cplusplus
}
#endif
#endif /* RDOOM_H */
