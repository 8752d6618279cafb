/* ORACLE (test infrastructure only) -- scalar CPU restatement of the reference's per-frame draw.
 *
 * What is restated (reference has no CPU rasteriser; glium/OpenGL does this work):
 *   - vertex stage      assets/shaders/static.vert:25-45
 *   - fixed function    engine/src/renderer.rs:49-57 (depth LESS + write, cull clockwise),
 *                       engine/src/window.rs:12,40-44 (24-bit depth, clear depth 1.0),
 *                       engine/src/meshes.rs:97-106 (TrianglesList, u32 indices),
 *                       draw order game/src/level.rs:443-496 (earlier primitive wins depth ties)
 *   - fragment stage    assets/shaders/static.frag:18-28 with samplers game_shaders.rs:133-142
 *                       (palette CLAMP/NEAREST) and :395-404 (atlas REPEAT/NEAREST)
 *   - sky               assets/shaders/sky.vert:9-16, sky.frag:12-26 (KIND_SKY)
 *   - decor billboards  assets/shaders/sprite.vert:23-47, sprite.frag:15-27 (KIND_DECOR), vertex layout
 *                       game/src/vertex.rs:30-40, quad emission game/src/level.rs:764-793
 *
 * OpenGL leaves sub-ulp behaviour to the driver, so "the arithmetic" is pinned HERE and in
 * DESIGN.md section "Raster arithmetic" -- binary32 (the triangle set-up S3..S5: binary64), explicit operation order, fmaf only where
 * written, no contraction (build with -ffp-contract=off).  The HIP kernels are written
 * independently against that text; this file is the checker.  It is never linked into the product.
 *
 * PARITY PIN STATUS: pinned against GL readbacks of the reference's own six shaders, executed headless by SwiftShader
 * (tests/gl_readback.py, fixtures tests/golden/gl_readback/, tests/test_gl_readback.py): 99.5 % of 147 M pixels in 195
 * frames identical, every other pixel explained by a discontinuity GL leaves to the implementation (tests/gl_census.py);
 * plus analytic KATs (tests/test_kat_analytic.py) and golden digests (tests/golden/).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KIND_FLAT 0u
#define KIND_WALL 1u
#define KIND_DECOR 2u
#define KIND_SKY 3u

typedef struct {
  float a_pos[3];
  float a_atlas_uv[2];
  float a_tile_uv[2];
  float a_tile_size[2];
  float a_scroll_rate;
  float a_row_height;
  uint8_t a_num_frames;
  uint8_t a_light;
  uint8_t pad[2];
} StaticVertex; /* game/src/vertex.rs:5-16, 48 bytes */

typedef struct {
  float a_pos[3];
  float a_atlas_uv[2];
  float a_tile_uv[2];
  float a_tile_size[2];
  float a_local_x;
  uint8_t a_num_frames;
  uint8_t a_light;
  uint8_t pad[2];
} SpriteVertex; /* game/src/vertex.rs:30-40, 44 bytes */

typedef struct {
  uint32_t kind, object_id, first_index, index_count;
} Draw;

typedef struct {
  const StaticVertex *static_verts;
  const uint32_t *static_indices;
  const float *sky_verts; /* xyz triples */
  const uint32_t *sky_indices;
  const Draw *draws;
  uint32_t n_draws;
  const uint8_t *flat_atlas;
  uint32_t flat_w, flat_h;
  const uint16_t *wall_atlas;
  uint32_t wall_w, wall_h;
  const uint16_t *sky_tex;
  uint32_t sky_w, sky_h;
  float sky_band;
  const uint8_t *colormap; /* 32*256 */
  const SpriteVertex *decor_verts;
  const uint32_t *decor_indices;
  const uint16_t *decor_atlas;
  uint32_t decor_w, decor_h;
} OracleLevel;

typedef struct {
  float e[3][3];  /* edge functions  e_i = fma(A,px,fma(B,py,C)) */
  float zp[3];    /* window depth plane */
  float wp[3];    /* 1/w plane */
  float up[3], vp[3]; /* u/w, v/w planes */
  int tl[3];
  int x0, y0, x1, y1; /* inclusive pixel bbox */
  uint32_t kind;
  float atlas_u, atlas_v, size_x, size_y, light;
} Setup;

static float glsl_mod(float x, float y) { return x - y * floorf(x / y); }

/* static.vert:27-39 : animation-frame atlas offset (flat varying) */
static void atlas_uv_at(const StaticVertex *v, float time, float atlas_w, float *au, float *av) {
  if (v->a_num_frames == 1) {
    *au = v->a_atlas_uv[0];
    *av = v->a_atlas_uv[1];
    return;
  }
  const float anim_fps = 8.0f / 35.0f;
  float frame_index = time / anim_fps;
  frame_index = floorf(glsl_mod(frame_index, (float)v->a_num_frames));
  float atlas_u = v->a_atlas_uv[0] + frame_index * v->a_tile_size[0];
  float n_rows_down = ceilf((atlas_u + v->a_tile_size[0]) / atlas_w) - 1.0f;
  atlas_u = atlas_u + glsl_mod(atlas_w - v->a_atlas_uv[0], v->a_tile_size[0]) * n_rows_down;
  *au = atlas_u;
  *av = v->a_atlas_uv[1] + n_rows_down * v->a_row_height;
}

static void mat_mul(const float *p, const float *m, float *pm) { /* column-major, (P*M)[c][r] */
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++)
      pm[c * 4 + r] = ((p[0 * 4 + r] * m[c * 4 + 0] + p[1 * 4 + r] * m[c * 4 + 1]) + p[2 * 4 + r] * m[c * 4 + 2]) +
                      p[3 * 4 + r] * m[c * 4 + 3];
}

static void xform(const float *pm, const float *pos, float *clip) {
  for (int r = 0; r < 4; r++)
    clip[r] = fmaf(pm[8 + r], pos[2], fmaf(pm[4 + r], pos[1], fmaf(pm[0 + r], pos[0], pm[12 + r])));
}

/* Triangle setup: DESIGN.md "Raster arithmetic" steps S1..S6.  Returns 0 if culled. */
static int setup_tri(const float clip[3][4], const float u[3], const float v[3], int width, int height, float zk,
                     Setup *s) {
  if (clip[0][3] <= 0.0f && clip[1][3] <= 0.0f && clip[2][3] <= 0.0f) return 0;
  const float hw = 0.5f * (float)width, hh = 0.5f * (float)height;
  float xw[3], yw[3], w[3];
  for (int i = 0; i < 3; i++) {
    xw[i] = (clip[i][0] + clip[i][3]) * hw;
    yw[i] = (clip[i][1] + clip[i][3]) * hh;
    w[i] = clip[i][3];
  }
  /* S3..S5 in BINARY64 on the binary32 inputs (round 6): every product of two binary32 values is exact in binary64, so an edge
   * coefficient is ONE rounded difference -- and still exactly the negative of the neighbouring triangle's across a shared edge --,
   * and the determinant and the plane numerators no longer lose their leading digits where a triangle is thin on the screen (found
   * by the census against Mesa: the binary32 set-up put u/w, v/w, 1/w up to 3 texels / 0.1 % off there).  Each operation below is one
   * IEEE binary64 operation (no contraction); the stored coefficients are those values rounded to binary32. */
  double ed[3][3];
  for (int i = 0; i < 3; i++) {
    int j = (i + 1) % 3, k = (i + 2) % 3;
    ed[i][0] = (double)yw[j] * (double)w[k] - (double)yw[k] * (double)w[j];
    ed[i][1] = (double)xw[k] * (double)w[j] - (double)xw[j] * (double)w[k];
    ed[i][2] = (double)xw[j] * (double)yw[k] - (double)xw[k] * (double)yw[j];
    for (int c = 0; c < 3; c++) s->e[i][c] = (float)ed[i][c];
    s->tl[i] = (s->e[i][0] > 0.0f) || (s->e[i][0] == 0.0f && s->e[i][1] > 0.0f);
  }
  const double det = (double)w[0] * ed[0][2] + ((double)yw[0] * ed[0][1] + (double)xw[0] * ed[0][0]);
  if (!(det > 0.0)) return 0; /* cull clockwise + degenerate (renderer.rs:55) */
  /* S5: interpolate the residual Z - zk * W (zk = P[2][2] / P[2][3]; the constant P[3][2] for a perspective matrix),
   * not Z itself: the part zk * W interpolates to zk exactly, so rounding in the edge functions no longer leaks
   * |Z| ~ |W| into window depth (found by the GL-readback census: far slivers lost to surfaces behind them) */
  float rz[3];
  for (int i = 0; i < 3; i++) rz[i] = fmaf(-zk, clip[i][3], clip[i][2]);
  for (int c = 0; c < 3; c++) {
    const double nz = (double)rz[2] * ed[2][c] + ((double)rz[1] * ed[1][c] + (double)rz[0] * ed[0][c]);
    const double n1 = (ed[0][c] + ed[1][c]) + ed[2][c];
    const double nu = (double)u[2] * ed[2][c] + ((double)u[1] * ed[1][c] + (double)u[0] * ed[0][c]);
    const double nv = (double)v[2] * ed[2][c] + ((double)v[1] * ed[1][c] + (double)v[0] * ed[0][c]);
    const double zp = 0.5 * (nz / det);
    s->zp[c] = (float)(c == 2 ? zp + (0.5 * (double)zk + 0.5) : zp);
    s->wp[c] = (float)(n1 / det);
    s->up[c] = (float)(nu / det);
    s->vp[c] = (float)(nv / det);
  }
  /* bbox: part of the coverage definition */
  float wmin = fminf(w[0], fminf(w[1], w[2]));
  s->x0 = 0;
  s->y0 = 0;
  s->x1 = width - 1;
  s->y1 = height - 1;
  if (wmin >= 1e-5f) {
    float sx[3], sy[3];
    for (int i = 0; i < 3; i++) {
      sx[i] = xw[i] / w[i];
      sy[i] = yw[i] / w[i];
    }
    float fx0 = floorf(fminf(sx[0], fminf(sx[1], sx[2]))) - 1.0f;
    float fx1 = ceilf(fmaxf(sx[0], fmaxf(sx[1], sx[2]))) + 1.0f;
    float fy0 = floorf(fminf(sy[0], fminf(sy[1], sy[2]))) - 1.0f;
    float fy1 = ceilf(fmaxf(sy[0], fmaxf(sy[1], sy[2]))) + 1.0f;
    if (!(fx0 <= (float)(width - 1) && fx1 >= 0.0f && fy0 <= (float)(height - 1) && fy1 >= 0.0f)) return 0;
    s->x0 = (int)fmaxf(fx0, 0.0f);
    s->y0 = (int)fmaxf(fy0, 0.0f);
    s->x1 = (int)fminf(fx1, (float)(width - 1));
    s->y1 = (int)fminf(fy1, (float)(height - 1));
  }
  return 1;
}

static float plane(const float *p, float px, float py) { return fmaf(p[0], px, fmaf(p[1], py, p[2])); }

/* static.frag:19-20 texel fetch.  Returns the raw texel (u16; flats are promoted, never transparent). */
static uint32_t texel_at(const OracleLevel *L, const Setup *s, float tu, float tv);
static uint32_t fetch_texel(const OracleLevel *L, const Setup *s, float px, float py, float *dist, float *tuv) {
  float rw = plane(s->wp, px, py);
  float w = 1.0f / rw;
  float tu = plane(s->up, px, py) * w;
  float tv = plane(s->vp, px, py) * w;
  *dist = w;
  if (tuv) tuv[0] = tu, tuv[1] = tv;
  return texel_at(L, s, tu, tv);
}

/* static.frag:19-20 / sprite.frag:19 on given v_tile_uv: F2 (mod + atlas offset), F3 (REPEAT + NEAREST) */
static uint32_t texel_at(const OracleLevel *L, const Setup *s, float tu, float tv) {
  float uvx = glsl_mod(tu, s->size_x) + s->atlas_u;
  float uvy = glsl_mod(tv, s->size_y) + s->atlas_v;
  int ix = (int)floorf(uvx), iy = (int)floorf(uvy);
  if (s->kind == KIND_FLAT) {
    return L->flat_atlas[(size_t)(iy & (int)(L->flat_h - 1)) * L->flat_w + (size_t)(ix & (int)(L->flat_w - 1))];
  }
  if (s->kind == KIND_DECOR) {
    return L->decor_atlas[(size_t)(iy & (int)(L->decor_h - 1)) * L->decor_w + (size_t)(ix & (int)(L->decor_w - 1))];
  }
  return L->wall_atlas[(size_t)(iy & (int)(L->wall_h - 1)) * L->wall_w + (size_t)(ix & (int)(L->wall_w - 1))];
}

/* static.frag:24-26 */
static uint8_t shade(const OracleLevel *L, uint32_t idx, float v_light, float dist) {
  float dist_term = fminf(1.0f, 1.0f - 0.9f / (dist + 0.9f));
  float light = v_light * 2.0f - dist_term;
  float t = (1.0f - light) * 32.0f;
  int row = t < 0.0f ? 0 : (t >= 32.0f ? 31 : (int)floorf(t));
  return L->colormap[row * 256 + (int)idx];
}

/* sprite.frag:22-25: DIST_SCALE = 1.0, light = min(v_light, v_light * LIGHT_SCALE - dist_term) */
static uint8_t shade_decor(const OracleLevel *L, uint32_t idx, float v_light, float dist) {
  float dist_term = fminf(1.0f, 1.0f - 1.0f / (dist + 1.0f));
  float light = fminf(v_light, v_light * 2.0f - dist_term);
  float t = (1.0f - light) * 32.0f;
  int row = t < 0.0f ? 0 : (t >= 32.0f ? 31 : (int)floorf(t));
  return L->colormap[row * 256 + (int)idx];
}

/* sprite.vert:27-39: animation-frame atlas offset; rows advance by a_tile_size.y (not a row height) */
static void sprite_atlas_uv_at(const SpriteVertex *v, float time, float atlas_w, float *au, float *av) {
  if (v->a_num_frames == 1) {
    *au = v->a_atlas_uv[0];
    *av = v->a_atlas_uv[1];
    return;
  }
  const float anim_fps = 8.0f / 35.0f;
  float frame_index = time / anim_fps;
  frame_index = floorf(glsl_mod(frame_index, (float)v->a_num_frames));
  float atlas_u = v->a_atlas_uv[0] + frame_index * v->a_tile_size[0];
  float n_rows_down = ceilf((atlas_u + v->a_tile_size[0]) / atlas_w) - 1.0f;
  atlas_u = atlas_u + glsl_mod(atlas_w - v->a_atlas_uv[0], v->a_tile_size[0]) * n_rows_down;
  *au = atlas_u;
  *av = v->a_atlas_uv[1] + n_rows_down * v->a_tile_size[1];
}

/* sprite.vert:41-46: pos = a_pos + right * a_local_x with right = row 0 of the modelview;
 * projected = u_projection * (u_modelview * vec4(pos, 1)) -- NOT (P*M)*v as in static.vert.
 * Pinned arithmetic: p_i = fma(right_i, local_x, a_pos_i); eye and clip as fma chains. */
static void xform_decor(const float *m, const float *p, const SpriteVertex *v, float *clip) {
  float pos[3], eye[4];
  for (int i = 0; i < 3; i++) pos[i] = fmaf(m[4 * i], v->a_local_x, v->a_pos[i]);
  for (int r = 0; r < 4; r++)
    eye[r] = fmaf(m[8 + r], pos[2], fmaf(m[4 + r], pos[1], fmaf(m[0 + r], pos[0], m[12 + r])));
  for (int r = 0; r < 4; r++)
    clip[r] = fmaf(p[12 + r], eye[3], fmaf(p[8 + r], eye[2], fmaf(p[4 + r], eye[1], p[0 + r] * eye[0])));
}

static uint8_t sky_texel_colour(const OracleLevel *L, float uvx, float uvy);
/* sky.frag:12-26.  v_p = clip position interpolated (x/w, y/w are NDC), v_r flat per pose. */
static uint8_t shade_sky(const OracleLevel *L, float px, float py, int width, int height, const float *vr) {
  float ndc_x = px / (0.5f * (float)width) - 1.0f;
  float ndc_y = py / (0.5f * (float)height) - 1.0f;
  float uvx = ndc_x;
  float uvy = -ndc_y;
  uvx = uvx - 4.0f * vr[0] / 3.14159265358f;
  uvy = (uvy + 1.0f) + vr[1];
  float band = L->sky_band;
  if (uvy < 0.0f) {
    uvy = fabsf(glsl_mod(-uvy + band, band * 2.0f) - band);
  } else if (uvy >= 2.0f) {
    uvy = fabsf(glsl_mod((uvy - 2.0f) + band, band * 2.0f) - band);
  } else if (uvy >= 1.0f) {
    uvy = 1.0f - uvy;
  }
  return sky_texel_colour(L, uvx, uvy);
}

/* sky.frag:24-25 on the folded uv: REPEAT + NEAREST on normalised coordinates, palette row 0 */
static uint8_t sky_texel_colour(const OracleLevel *L, float uvx, float uvy) {
  float fx = uvx - floorf(uvx), fy = uvy - floorf(uvy);
  int ix = (int)floorf(fx * (float)L->sky_w), iy = (int)floorf(fy * (float)L->sky_h);
  if (ix >= (int)L->sky_w) ix = (int)L->sky_w - 1;
  if (iy >= (int)L->sky_h) iy = (int)L->sky_h - 1;
  uint32_t texel = L->sky_tex[(size_t)iy * L->sky_w + (size_t)ix];
  return L->colormap[texel & 0xFF]; /* palette row v = 0 -> colormap 0 */
}

#define NO_PRIM 0xFFFFFFFFu

/* Renders one pose.  out_fb: height*width bytes, row 0 = bottom (glReadPixels order).
 * out_prim (optional): winning primitive id per pixel or NO_PRIM.  kinds_mask: bit k enables KIND k. */
/* object_modelviews (optional): n_objects matrices, the u_modelview the reference sets for the draws of each
 * object = view o model transform (engine/src/renderer.rs:120-132; doors / lifts move their object's transform,
 * game/src/level.rs:203-255).  NULL = every object at identity: u_modelview = modelview for all draws. */
static int render_with_scratch(const OracleLevel *L, const float *modelview, const float *projection, float time,
                               const uint8_t *lights, int width, int height, uint32_t kinds_mask, uint8_t *out_fb,
                               uint32_t *out_prim, uint32_t *depth, uint32_t *prim, const float *object_modelviews,
                               uint32_t n_objects, float *out_var) {
  size_t npx = (size_t)width * (size_t)height;
  for (size_t i = 0; i < npx; i++) {
    depth[i] = 0xFFFFFFFFu;
    prim[i] = NO_PRIM;
  }
  memset(out_fb, 0, npx);
  const float zk = projection[11] != 0.0f ? projection[10] / projection[11] : 0.0f;
  uint32_t prim_id = 0;
  for (uint32_t d = 0; d < L->n_draws; d++) {
    const Draw *dr = &L->draws[d];
    uint32_t ntri = dr->index_count / 3;
    if (object_modelviews && dr->object_id < n_objects) modelview = object_modelviews + 16 * (size_t)dr->object_id;
    float pm[16];
    mat_mul(projection, modelview, pm);
    /* sky.vert:10-12 */
    float vr[2];
    vr[0] = atan2f(pm[8], pm[10]);
    vr[1] = pm[9] / pm[11];
    if (!((kinds_mask >> dr->kind) & 1u) || dr->kind > KIND_SKY) {
      prim_id += ntri;
      continue;
    }
    for (uint32_t t = 0; t < ntri; t++, prim_id++) {
      float clip[3][4], u[3] = {0, 0, 0}, v[3] = {0, 0, 0};
      Setup s;
      s.kind = dr->kind;
      if (dr->kind == KIND_SKY) {
        for (int i = 0; i < 3; i++) xform(pm, &L->sky_verts[3 * L->sky_indices[dr->first_index + 3 * t + i]], clip[i]);
      } else if (dr->kind == KIND_DECOR) {
        const SpriteVertex *vv[3];
        for (int i = 0; i < 3; i++) {
          vv[i] = &L->decor_verts[L->decor_indices[dr->first_index + 3 * t + i]];
          xform_decor(modelview, projection, vv[i], clip[i]);
          u[i] = vv[i]->a_tile_uv[0]; /* sprite.vert:24: no scroll */
          v[i] = vv[i]->a_tile_uv[1];
        }
        const SpriteVertex *pv = vv[2];
        sprite_atlas_uv_at(pv, time, (float)L->decor_w, &s.atlas_u, &s.atlas_v);
        s.size_x = pv->a_tile_size[0];
        s.size_y = pv->a_tile_size[1];
        s.light = (float)lights[pv->a_light] / 255.0f;
      } else {
        const StaticVertex *vv[3];
        for (int i = 0; i < 3; i++) {
          vv[i] = &L->static_verts[L->static_indices[dr->first_index + 3 * t + i]];
          xform(pm, vv[i]->a_pos, clip[i]);
          u[i] = vv[i]->a_tile_uv[0] + time * vv[i]->a_scroll_rate; /* static.vert:26 */
          v[i] = vv[i]->a_tile_uv[1];
        }
        const StaticVertex *pv = vv[2]; /* flat varyings: provoking (last) vertex */
        float aw = dr->kind == KIND_FLAT ? (float)L->flat_w : (float)L->wall_w;
        atlas_uv_at(pv, time, aw, &s.atlas_u, &s.atlas_v);
        s.size_x = pv->a_tile_size[0];
        s.size_y = pv->a_tile_size[1];
        s.light = (float)lights[pv->a_light] / 255.0f; /* static.vert:43 texelFetch of R8 unorm */
      }
      if (!setup_tri(clip, u, v, width, height, zk, &s)) continue;
      for (int iy = s.y0; iy <= s.y1; iy++) {
        float py = (float)iy + 0.5f;
        for (int ix = s.x0; ix <= s.x1; ix++) {
          float px = (float)ix + 0.5f;
          int inside = 1;
          for (int i = 0; i < 3 && inside; i++) {
            float e = plane(s.e[i], px, py);
            inside = (e > 0.0f) || (e == 0.0f && s.tl[i]);
          }
          if (!inside) continue;
          float zw = plane(s.zp, px, py);
          if (!(zw >= 0.0f && zw <= 1.0f)) continue;
          float rw = plane(s.wp, px, py);
          if (!(rw > 0.0f)) continue;
          uint32_t d24 = (uint32_t)fmaf(zw, 16777215.0f, 0.5f);
          size_t o = (size_t)iy * (size_t)width + (size_t)ix;
          if (!(d24 < depth[o])) continue; /* LESS: the earlier primitive keeps ties */
          uint8_t colour;
          float var[3] = {0.0f, 0.0f, 0.0f};
          if (dr->kind == KIND_SKY) {
            colour = shade_sky(L, px, py, width, height, vr);
          } else {
            float dist;
            float tuv[2];
            uint32_t texel = fetch_texel(L, &s, px, py, &dist, tuv);
            if (dr->kind != KIND_FLAT && (texel & 0x8000u)) continue; /* static.frag:21 / sprite.frag:20 discard */
            colour = dr->kind == KIND_DECOR ? shade_decor(L, texel & 0xFFu, s.light, dist)
                                            : shade(L, texel & 0xFFu, s.light, dist);
            var[0] = tuv[0], var[1] = tuv[1], var[2] = dist;
          }
          depth[o] = d24;
          prim[o] = prim_id;
          out_fb[o] = colour;
          if (out_var && dr->kind != KIND_SKY) out_var[3 * o] = var[0], out_var[3 * o + 1] = var[1], out_var[3 * o + 2] = var[2];
        }
      }
    }
  }
  if (out_prim) memcpy(out_prim, prim, npx * sizeof(uint32_t));
  return 0;
}

int oracle_render(const OracleLevel *L, const float *modelview, const float *projection, float time,
                  const uint8_t *lights, int width, int height, uint32_t kinds_mask, uint8_t *out_fb,
                  uint32_t *out_prim) {
  size_t npx = (size_t)width * (size_t)height;
  uint32_t *depth = (uint32_t *)malloc(npx * sizeof(uint32_t));
  uint32_t *prim = (uint32_t *)malloc(npx * sizeof(uint32_t));
  int rc = -1;
  if (depth && prim)
    rc = render_with_scratch(L, modelview, projection, time, lights, width, height, kinds_mask, out_fb, out_prim, depth,
                             prim, 0, 0, 0);
  free(depth);
  free(prim);
  return rc;
}

int oracle_render_objects(const OracleLevel *L, const float *modelview, const float *projection, float time,
                          const uint8_t *lights, int width, int height, uint32_t kinds_mask, uint8_t *out_fb,
                          uint32_t *out_prim, const float *object_modelviews, uint32_t n_objects) {
  size_t npx = (size_t)width * (size_t)height;
  uint32_t *depth = (uint32_t *)malloc(npx * sizeof(uint32_t));
  uint32_t *prim = (uint32_t *)malloc(npx * sizeof(uint32_t));
  int rc = -1;
  if (depth && prim)
    rc = render_with_scratch(L, modelview, projection, time, lights, width, height, kinds_mask, out_fb, out_prim, depth,
                             prim, object_modelviews, n_objects, 0);
  free(depth);
  free(prim);
  return rc;
}

/* Renders a batch of poses with `threads`-way pose parallelism left to the caller (ctypes releases the
 * GIL); poses: n * (16 modelview + 16 projection + 1 time) floats; lights: n * 256 bytes. */
int oracle_render_batch(const OracleLevel *L, const float *poses, const uint8_t *lights, int n, int width, int height,
                        uint32_t kinds_mask, uint8_t *out_fb) {
  size_t npx = (size_t)width * (size_t)height;
  uint32_t *depth = (uint32_t *)malloc(npx * sizeof(uint32_t)); /* one scratch pair per caller thread */
  uint32_t *prim = (uint32_t *)malloc(npx * sizeof(uint32_t));
  int rc = (depth && prim) ? 0 : -1;
  for (int i = 0; i < n && !rc; i++) {
    const float *p = poses + (size_t)i * 33;
    rc = render_with_scratch(L, p, p + 16, p[32], lights + (size_t)i * 256, width, height, kinds_mask,
                             out_fb + (size_t)i * npx, 0, depth, prim, 0, 0, 0);
  }
  free(depth);
  free(prim);
  return rc;
}

/* Census support (tests/gl_census.py): the frame plus, per pixel, the varyings the winning fragment was shaded from --
 * v_tile_uv.x, v_tile_uv.y, v_dist as THIS arithmetic evaluates them (sky and background pixels: zeros). */
int oracle_render_varyings(const OracleLevel *L, const float *modelview, const float *projection, float time,
                           const uint8_t *lights, int width, int height, uint32_t kinds_mask, uint8_t *out_fb,
                           uint32_t *out_prim, float *out_var) {
  size_t npx = (size_t)width * (size_t)height;
  uint32_t *depth = (uint32_t *)malloc(npx * sizeof(uint32_t));
  uint32_t *prim = (uint32_t *)malloc(npx * sizeof(uint32_t));
  int rc = -1;
  if (depth && prim) {
    memset(out_var, 0, npx * 3 * sizeof(float));
    rc = render_with_scratch(L, modelview, projection, time, lights, width, height, kinds_mask, out_fb, out_prim, depth,
                             prim, 0, 0, out_var);
  }
  free(depth);
  free(prim);
  return rc;
}

/* Zero-tolerance pin of the fragment stage (tests/gl_census.py: fragment_exact): THIS file's binary32 fragment code --
 * F2..F6 of DESIGN.md, i.e. static.frag:18-28, sprite.frag:15-27, sky.frag:24-25 -- applied to varyings that come from
 * OUTSIDE (the ones the reference's own shaders interpolated under SwiftShader, read back by an auxiliary pass) for the
 * primitive that won there.  in_prim: primitive id per pixel (NO_PRIM: nothing drawn); in_var: per pixel
 * (v_tile_uv.x, v_tile_uv.y, v_dist), or the folded (uv.x, uv.y, -1) of sky.frag:23.  out: palette index 0..255,
 * 0x100 = the fragment would have been discarded (transparent texel), 0xFFFF = no primitive. */
int oracle_shade_varyings(const OracleLevel *L, float time, const uint8_t *lights, int width, int height,
                          const uint32_t *in_prim, const float *in_var, uint16_t *out) {
  size_t npx = (size_t)width * (size_t)height;
  uint32_t total = 0;
  for (uint32_t d = 0; d < L->n_draws; d++) total += L->draws[d].index_count / 3;
  uint32_t *draw_of = (uint32_t *)malloc(((size_t)total + 1) * sizeof(uint32_t));
  if (!draw_of) return -1;
  uint32_t *first_of = (uint32_t *)malloc(((size_t)L->n_draws + 1) * sizeof(uint32_t));
  if (!first_of) {
    free(draw_of);
    return -1;
  }
  uint32_t at = 0;
  for (uint32_t d = 0; d < L->n_draws; d++) {
    first_of[d] = at;
    for (uint32_t t = 0; t < L->draws[d].index_count / 3; t++) draw_of[at++] = d;
  }
  for (size_t o = 0; o < npx; o++) {
    uint32_t pid = in_prim[o];
    if (pid == NO_PRIM || pid >= total) {
      out[o] = 0xFFFFu;
      continue;
    }
    const Draw *dr = &L->draws[draw_of[pid]];
    uint32_t t = pid - first_of[draw_of[pid]];
    const float *var = in_var + 3 * o;
    if (dr->kind == KIND_SKY) {
      out[o] = sky_texel_colour(L, var[0], var[1]);
      continue;
    }
    Setup s;
    memset(&s, 0, sizeof s);
    s.kind = dr->kind;
    if (dr->kind == KIND_DECOR) {
      const SpriteVertex *pv = &L->decor_verts[L->decor_indices[dr->first_index + 3 * t + 2]];
      sprite_atlas_uv_at(pv, time, (float)L->decor_w, &s.atlas_u, &s.atlas_v);
      s.size_x = pv->a_tile_size[0], s.size_y = pv->a_tile_size[1];
      s.light = (float)lights[pv->a_light] / 255.0f;
    } else {
      const StaticVertex *pv = &L->static_verts[L->static_indices[dr->first_index + 3 * t + 2]];
      atlas_uv_at(pv, time, dr->kind == KIND_FLAT ? (float)L->flat_w : (float)L->wall_w, &s.atlas_u, &s.atlas_v);
      s.size_x = pv->a_tile_size[0], s.size_y = pv->a_tile_size[1];
      s.light = (float)lights[pv->a_light] / 255.0f;
    }
    uint32_t texel = texel_at(L, &s, var[0], var[1]);
    if (dr->kind != KIND_FLAT && (texel & 0x8000u)) {
      out[o] = 0x100u;
      continue;
    }
    out[o] = dr->kind == KIND_DECOR ? shade_decor(L, texel & 0xFFu, s.light, var[2]) : shade(L, texel & 0xFFu, s.light, var[2]);
  }
  free(first_of);
  free(draw_of);
  return 0;
}
