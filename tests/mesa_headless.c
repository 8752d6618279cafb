/* Headless OpenGL (desktop, core profile) on Mesa's software rasteriser (llvmpipe) WITHOUT an X server, EGL or OSMesa: this image
 * ships Mesa's DRI driver (/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so) but nothing that could hand it a window.  The driver is
 * opened the way Mesa's own GLX client does for its "drisw" path (an X drawable whose pixels the loader moves): through the
 * driver's extension table -- DRI_Core + DRI_SWRast -- with a loader extension (DRI_SWRastLoader) whose put/get-image callbacks
 * do nothing, because every frame here is rendered into a framebuffer object and read back with glReadPixels.
 *
 * TEST INFRASTRUCTURE (tests/gl_readback.py, backend "mesa"): a SECOND, independent GL implementation that executes the
 * reference's six GLSL programs -- unpatched, `samplerBuffer` included, with the `#version 140` line the engine itself prepends
 * (engine/src/platform.rs:5, engine/src/shaders.rs:45) -- next to SwiftShader's OpenGL ES 3.0.  Never linked into the product.
 * Build: gcc -shared -fPIC -O1 -o tests/_build/libmesa_headless.so tests/mesa_headless.c -ldl   (tests/gl_readback.py does it). */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <GL/internal/dri_interface.h>

static int g_w = 16, g_h = 16;
static void *g_driver, *g_glapi;
static const __DRIcoreExtension *g_core;
static const __DRIswrastExtension *g_swrast;
static __DRIscreen *g_screen;
static __DRIcontext *g_ctx;
static __DRIdrawable *g_draw;
static void *(*g_get_proc)(const char *);
static char g_err[256];

static void get_drawable_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *lp) {
  (void)d, (void)lp;
  *x = 0, *y = 0, *w = g_w, *h = g_h;
}
static void put_image(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *lp) {
  (void)d, (void)op, (void)x, (void)y, (void)w, (void)h, (void)data, (void)lp; /* the window system framebuffer is never looked at */
}
static void get_image(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *lp) {
  (void)d, (void)x, (void)y, (void)lp;
  memset(data, 0, (size_t)w * (size_t)h * 4u);
}
static void put_image2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *lp) {
  (void)d, (void)op, (void)x, (void)y, (void)w, (void)h, (void)stride, (void)data, (void)lp;
}
static void get_image2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *lp) {
  (void)d, (void)x, (void)y, (void)w, (void)lp;
  memset(data, 0, (size_t)stride * (size_t)h);
}
static const __DRIswrastLoaderExtension g_loader = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = get_drawable_info,
    .putImage = put_image,
    .getImage = get_image,
    .putImage2 = put_image2,
    .getImage2 = get_image2,
};
static const __DRIextension *g_loader_exts[] = {&g_loader.base, NULL};

const char *mesa_headless_error(void) { return g_err; }

/* 0 on success.  driver_path: swrast_dri.so; a core-profile context of at least major.minor is made current on this thread. */
int mesa_headless_init(const char *driver_path, const char *glapi_path, int major, int minor) {
  if (g_ctx) return 0;
  g_glapi = dlopen(glapi_path, RTLD_NOW | RTLD_GLOBAL);
  if (!g_glapi) return snprintf(g_err, sizeof g_err, "dlopen %s: %s", glapi_path, dlerror()), 1;
  g_get_proc = (void *(*)(const char *))dlsym(g_glapi, "_glapi_get_proc_address");
  if (!g_get_proc) return snprintf(g_err, sizeof g_err, "_glapi_get_proc_address not exported"), 2;
  g_driver = dlopen(driver_path, RTLD_NOW | RTLD_GLOBAL);
  if (!g_driver) return snprintf(g_err, sizeof g_err, "dlopen %s: %s", driver_path, dlerror()), 3;
  const __DRIextension **(*get_exts)(void) = (const __DRIextension **(*)(void))dlsym(g_driver, "__driDriverGetExtensions_swrast");
  if (!get_exts) return snprintf(g_err, sizeof g_err, "__driDriverGetExtensions_swrast not exported"), 4;
  const __DRIextension **exts = get_exts();
  for (int i = 0; exts && exts[i]; i++) {
    if (strcmp(exts[i]->name, __DRI_CORE) == 0) g_core = (const __DRIcoreExtension *)exts[i];
    if (strcmp(exts[i]->name, __DRI_SWRAST) == 0) g_swrast = (const __DRIswrastExtension *)exts[i];
  }
  if (!g_core || !g_swrast || g_swrast->base.version < 4)
    return snprintf(g_err, sizeof g_err, "driver lacks DRI_Core / DRI_SWRast >= 4"), 5;
  const __DRIconfig **configs = NULL;
  g_screen = g_swrast->createNewScreen2(0, g_loader_exts, exts, &configs, NULL);
  if (!g_screen || !configs) return snprintf(g_err, sizeof g_err, "createNewScreen2 failed"), 6;
  const __DRIconfig *pick = NULL;
  for (int i = 0; configs[i]; i++) { /* RGBA8888 + 24-bit depth (engine/src/window.rs:12), single buffered if there is one */
    unsigned r = 0, a = 0, z = 0, db = 1;
    g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_RED_SIZE, &r);
    g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_ALPHA_SIZE, &a);
    g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_DEPTH_SIZE, &z);
    g_core->getConfigAttrib(configs[i], __DRI_ATTRIB_DOUBLE_BUFFER, &db);
    if (r == 8 && z == 24 && (!pick || (a == 8 && !db))) pick = configs[i];
  }
  if (!pick) pick = configs[0];
  g_draw = g_swrast->createNewDrawable(g_screen, pick, NULL);
  if (!g_draw) return snprintf(g_err, sizeof g_err, "createNewDrawable failed"), 7;
  const uint32_t attribs[4] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, (uint32_t)major, __DRI_CTX_ATTRIB_MINOR_VERSION, (uint32_t)minor};
  unsigned error = 0;
  g_ctx = g_swrast->createContextAttribs(g_screen, __DRI_API_OPENGL_CORE, pick, NULL, 2, attribs, &error, NULL);
  if (!g_ctx) return snprintf(g_err, sizeof g_err, "createContextAttribs(core %d.%d) failed: error %u", major, minor, error), 8;
  if (!g_core->bindContext(g_ctx, g_draw, g_draw)) return snprintf(g_err, sizeof g_err, "bindContext failed"), 9;
  return 0;
}

void *mesa_headless_proc(const char *name) { return g_get_proc ? g_get_proc(name) : NULL; }
