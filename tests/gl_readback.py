"""GL readback of the REFERENCE's own shaders -- the reference-executed pin of the GPU half (rows a17-a19, f1, f2).

The six GLSL programs of cristicbz/rust-doom (assets/shaders/{static,sky,sprite}.{vert,frag}) are loaded from the
reference checkout AT RUN TIME (never copied into this repository) and executed headless by the software GL that
ships in this image: SwiftShader (OpenGL ES 3.0, GLSL ES 3.00) inside the `kaleido` wheel, opened through ctypes/EGL.
Everything the reference hands to glium is reproduced as it hands it over:

* vertex/index buffers: the `StaticVertex` / `SkyVertex` / `SpriteVertex` streams and u32 index lists of
  game::level::Builder::build (game/src/level.rs:424-496), TrianglesList (engine/src/meshes.rs:97-106);
* textures: flat atlas `U8` -> R8, wall / decor atlases and the sky texture `U8U8` -> RG8, REPEAT / NEAREST
  (game/src/game_shaders.rs:389-453); palette = build_palette_texture(0, 0, 32): 256 x 32 `U8U8U8`, CLAMP / NEAREST
  (wad/src/tex.rs:137-166, game_shaders.rs:123-144); lights = 256 normalised u8 (game_shaders.rs:152-159);
* state: depth IfLess + write, CullClockwise (engine/src/renderer.rs:49-57), 24-bit depth (window.rs:12),
  clear (0.06, 0.07, 0.09, 0) / depth 1 (window.rs:40-44);
* draw order: per object flats -> walls -> decor -> sky (level.rs:443-496), one `u_modelview` per object
  (renderer.rs:120-132).

Only what GLSL ES 3.00 forces is patched in the shader text (`patch_shader`):
  1. the `#version` line the engine prepends (engine/src/shaders.rs:45) becomes `#version 300 es`;
  2. `precision mediump float;` (ignored by desktop GLSL 1.40, binding in ES) becomes `precision highp float;`, and the
     vertex stage's implicit highp is spelled out for int and samplers;
  3. `samplerBuffer` does not exist in ES 3.00: `u_lights` becomes a 256x1 R8 `sampler2D` read with
     `texelFetch(u_lights, ivec2(a_light, 0), 0)` -- the same normalised-u8 fetch.

Two auxiliary passes exist only to classify mismatching pixels (tests/gl_census.py); neither is compared with anything:
`mode='ids'` draws the same geometry de-indexed with one extra flat attribute and the colour write replaced by that
attribute, which reads back WHICH primitive SwiftShader's rasteriser + depth test chose (gl_PrimitiveID does not exist
in ES 3.00); `mode='varyings'` replaces the colour write by `vec3(v_tile_uv, v_dist)` into a float target, which reads
back the varyings SwiftShader interpolated -- GL leaves their precision to the implementation.

A SECOND GL IMPLEMENTATION (round 6): Mesa's llvmpipe, desktop OpenGL 4.5 core profile, opened without an X server through
the DRI driver's own extension table (tests/mesa_headless.c).  There NOTHING in the shader text is patched: the engine's
`#version 140` line (engine/src/platform.rs:5, engine/src/shaders.rs:45) is prepended as the engine prepends it, `u_lights` is
the `samplerBuffer` the reference declares (a GL_R8 buffer texture, as game_shaders.rs:152-159 creates it), `precision mediump
float;` stays (desktop GLSL ignores it).  `GLReference(level, backend='mesa')`; the auxiliary passes replace the colour write
the same way.  tests/test_gl_readback_mesa.py holds what is asserted about it.

Test infrastructure: imported by tests/golden/make_gl_readback.py (fixture generator) and tests/test_gl_readback.py.
"""
import ctypes
import os
import re

import numpy as np

SWIFTSHADER_DIR = os.environ.get(
    'RDOOM_SWIFTSHADER_DIR', '/usr/local/lib/python3.10/dist-packages/kaleido/executable/bin/swiftshader')
MESA_DRIVER = os.environ.get('RDOOM_MESA_SWRAST', '/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so')
MESA_GLAPI = os.environ.get('RDOOM_MESA_GLAPI', '/usr/lib/x86_64-linux-gnu/libglapi.so.0')
ENGINE_VERSION_LINE = '#version 140\n'   # engine/src/shaders.rs:45 with platform::GLSL_VERSION_STRING = "140" (engine/src/platform.rs:5)
GL_TEXTURE_BUFFER = 0x8C2A
REFERENCE_SHADERS = os.environ.get('RDOOM_REFERENCE_SHADERS', '/root/reference/assets/shaders')
CLEAR_RGB = (15, 18, 23)   # (0.06, 0.07, 0.09) as 8-bit UNORM (window.rs:42)
KIND_FLAT, KIND_WALL, KIND_DECOR, KIND_SKY = 0, 1, 2, 3
NO_PRIM = 0xFFFFFF

# --- GL / EGL constants -------------------------------------------------------------------------------------------
EGL_NONE, EGL_PBUFFER_BIT, EGL_OPENGL_ES3_BIT, EGL_OPENGL_ES_API = 0x3038, 1, 0x40, 0x30A0
EGL_SURFACE_TYPE, EGL_RENDERABLE_TYPE, EGL_DEPTH_SIZE = 0x3033, 0x3040, 0x3025
EGL_RED_SIZE, EGL_GREEN_SIZE, EGL_BLUE_SIZE, EGL_WIDTH, EGL_HEIGHT = 0x3024, 0x3023, 0x3022, 0x3057, 0x3056
EGL_CONTEXT_CLIENT_VERSION = 0x3098
GL_VERTEX_SHADER, GL_FRAGMENT_SHADER, GL_COMPILE_STATUS, GL_LINK_STATUS = 0x8B31, 0x8B30, 0x8B81, 0x8B82
GL_ARRAY_BUFFER, GL_ELEMENT_ARRAY_BUFFER, GL_STATIC_DRAW = 0x8892, 0x8893, 0x88E4
GL_TEXTURE_2D, GL_TEXTURE0 = 0x0DE1, 0x84C0
GL_TEXTURE_MIN_FILTER, GL_TEXTURE_MAG_FILTER, GL_TEXTURE_WRAP_S, GL_TEXTURE_WRAP_T = 0x2801, 0x2800, 0x2802, 0x2803
GL_NEAREST, GL_REPEAT, GL_CLAMP_TO_EDGE = 0x2600, 0x2901, 0x812F
GL_RGBA32F = 0x8814
GL_R8, GL_RG8, GL_RGB8, GL_RGBA8, GL_RED, GL_RG, GL_RGB, GL_RGBA = 0x8229, 0x822B, 0x8051, 0x8058, 0x1903, 0x8227, 0x1907, 0x1908
GL_UNSIGNED_BYTE, GL_UNSIGNED_INT, GL_FLOAT, GL_BYTE_T = 0x1401, 0x1405, 0x1406, 0x1400
GL_FRAMEBUFFER, GL_RENDERBUFFER, GL_COLOR_ATTACHMENT0, GL_DEPTH_ATTACHMENT = 0x8D40, 0x8D41, 0x8CE0, 0x8D00
GL_DEPTH_COMPONENT24, GL_DEPTH_COMPONENT, GL_FRAMEBUFFER_COMPLETE = 0x81A6, 0x1902, 0x8CD5
GL_DEPTH_TEST, GL_CULL_FACE, GL_LESS, GL_BACK, GL_CCW, GL_DITHER, GL_BLEND = 0x0B71, 0x0B44, 0x0201, 0x0405, 0x0901, 0x0BD0, 0x0BE2
GL_COLOR_BUFFER_BIT, GL_DEPTH_BUFFER_BIT, GL_TRIANGLES = 0x4000, 0x0100, 0x0004
GL_UNPACK_ALIGNMENT, GL_PACK_ALIGNMENT = 0x0CF5, 0x0D05


def available(backend='swiftshader'):
    if not os.path.exists(os.path.join(REFERENCE_SHADERS, 'static.frag')):
        return False
    if backend == 'mesa':
        return os.path.exists(MESA_DRIVER) and os.path.exists(MESA_GLAPI) and os.path.exists('/usr/include/GL/internal/dri_interface.h')
    return os.path.exists(os.path.join(SWIFTSHADER_DIR, 'libEGL.so'))


def patch_shader(text, stage, mode='colour', backend='swiftshader'):
    """The reference's shader text with only the GLSL ES 3.00 necessities changed (module docstring, 1-3); backend 'mesa'
    (desktop GL): the text AS IT IS behind the engine's own version line.
    mode 'ids' / 'varyings': the auxiliary passes (the colour write is replaced, nothing else)."""
    if backend == 'mesa':
        head = ENGINE_VERSION_LINE
    else:
        text = text.replace('precision mediump float;', '')
        head = '#version 300 es\nprecision highp float;\nprecision highp int;\nprecision highp sampler2D;\n'
        text = text.replace('uniform samplerBuffer u_lights;', 'uniform sampler2D u_lights;')
        text = text.replace('texelFetch(u_lights, a_light)', 'texelFetch(u_lights, ivec2(a_light, 0), 0)')
    if mode == 'varyings' and stage == 'frag':   # auxiliary pass: the varyings SwiftShader interpolated at this pixel
        text, n = re.subn(r'color = texture\(u_palette, vec2\(palette_index\.r,[^;]*;', 'color = vec3(v_tile_uv, v_dist);', text)
        if n == 0:
            text, n = re.subn(r'color = texture\(u_palette, vec2\(palette_index, 0\.0\)\)\.rgb;', 'color = vec3(uv, -1.0);', text)
        assert n == 1
    if mode == 'ids':   # auxiliary pass: which primitive won (never used for the colour comparison)
        if stage == 'vert':
            text = text.replace('void main() {', 'in vec3 x_id;\nflat out vec3 y_id;\nvoid main() {\n    y_id = x_id;', 1)
        else:
            text = text.replace('void main() {', 'flat in vec3 y_id;\nvoid main() {', 1)
            text, n = re.subn(r'color = texture\(u_palette,[^;]*;', 'color = y_id;', text)
            assert n == 1
    return head + text


class _GL:
    """ctypes bindings for the handful of GLES3/EGL entry points used."""

    def __init__(self):
        mode = ctypes.RTLD_GLOBAL
        self.gles = ctypes.CDLL(os.path.join(SWIFTSHADER_DIR, 'libGLESv2.so'), mode=mode)
        self.egl = ctypes.CDLL(os.path.join(SWIFTSHADER_DIR, 'libEGL.so'), mode=mode)
        vp, ci, cu, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_float
        e, g = self.egl, self.gles
        e.eglGetDisplay.restype, e.eglGetDisplay.argtypes = vp, [vp]
        e.eglInitialize.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci)]
        e.eglChooseConfig.argtypes = [vp, vp, ctypes.POINTER(vp), ci, ctypes.POINTER(ci)]
        e.eglCreatePbufferSurface.restype, e.eglCreatePbufferSurface.argtypes = vp, [vp, vp, vp]
        e.eglCreateContext.restype, e.eglCreateContext.argtypes = vp, [vp, vp, vp, vp]
        e.eglMakeCurrent.argtypes = [vp, vp, vp, vp]
        g.glGetString.restype = ctypes.c_char_p
        g.glShaderSource.argtypes = [cu, ci, ctypes.POINTER(ctypes.c_char_p), vp]
        g.glGetShaderiv.argtypes = g.glGetProgramiv.argtypes = [cu, cu, ctypes.POINTER(ci)]
        g.glGetShaderInfoLog.argtypes = g.glGetProgramInfoLog.argtypes = [cu, ci, vp, ctypes.c_char_p]
        g.glGetUniformLocation.argtypes = g.glGetAttribLocation.argtypes = [cu, ctypes.c_char_p]
        g.glGetUniformLocation.restype = g.glGetAttribLocation.restype = ci
        g.glBufferData.argtypes = [cu, ctypes.c_ssize_t, vp, cu]
        g.glTexImage2D.argtypes = [cu, ci, ci, ci, ci, ci, cu, cu, vp]
        g.glVertexAttribPointer.argtypes = [cu, ci, cu, ctypes.c_ubyte, ci, vp]
        g.glVertexAttribIPointer.argtypes = [cu, ci, cu, ci, vp]
        g.glUniformMatrix4fv.argtypes = [ci, ci, ctypes.c_ubyte, vp]
        g.glUniform1f.argtypes = [ci, cf]
        g.glUniform2f.argtypes = [ci, cf, cf]
        g.glUniform1i.argtypes = [ci, ci]
        g.glClearColor.argtypes = [cf, cf, cf, cf]
        g.glClearDepthf.argtypes = [cf]
        g.glDrawElements.argtypes = [cu, ci, cu, vp]
        g.glReadPixels.argtypes = [ci, ci, ci, ci, cu, cu, vp]
        g.glGenBuffers.argtypes = g.glGenTextures.argtypes = g.glGenFramebuffers.argtypes = \
            g.glGenRenderbuffers.argtypes = g.glGenVertexArrays.argtypes = [ci, ctypes.POINTER(cu)]
        g.glDeleteBuffers.argtypes = g.glDeleteTextures.argtypes = g.glDeleteFramebuffers.argtypes = \
            g.glDeleteRenderbuffers.argtypes = [ci, ctypes.POINTER(cu)]
        dpy = e.eglGetDisplay(None)
        major, minor = ci(), ci()
        if not e.eglInitialize(dpy, major, minor):
            raise RuntimeError('eglInitialize failed')
        attrs = (ci * 13)(EGL_SURFACE_TYPE, EGL_PBUFFER_BIT, EGL_RENDERABLE_TYPE, EGL_OPENGL_ES3_BIT, EGL_DEPTH_SIZE, 24,
                          EGL_RED_SIZE, 8, EGL_GREEN_SIZE, 8, EGL_BLUE_SIZE, 8, EGL_NONE)
        cfg, n = vp(), ci()
        if not e.eglChooseConfig(dpy, attrs, ctypes.byref(cfg), 1, n) or n.value < 1:
            raise RuntimeError('no EGL config with a 24-bit depth buffer')
        surf = e.eglCreatePbufferSurface(dpy, cfg, (ci * 5)(EGL_WIDTH, 16, EGL_HEIGHT, 16, EGL_NONE))
        e.eglBindAPI(EGL_OPENGL_ES_API)
        ctx = e.eglCreateContext(dpy, cfg, None, (ci * 3)(EGL_CONTEXT_CLIENT_VERSION, 3, EGL_NONE))
        if not ctx or not e.eglMakeCurrent(dpy, surf, surf, ctx):
            raise RuntimeError('could not create an OpenGL ES 3 context')
        self.version = g.glGetString(0x1F02).decode()
        self.glsl_version = g.glGetString(0x8B8C).decode()
        bits = ci()
        g.glGetIntegerv(0x0D50, ctypes.byref(bits))
        self.subpixel_bits = bits.value

    def gen(self, fn):
        v = ctypes.c_uint()
        fn(1, ctypes.byref(v))
        return v.value

    def check(self, where):
        err = self.gles.glGetError()
        if err:
            raise RuntimeError('GL error 0x%x at %s' % (err, where))


class _Namespace:
    pass


class _GLMesa:
    """Desktop OpenGL (core profile) on Mesa's llvmpipe through tests/mesa_headless.c: the same attribute names as _GL
    (`gles` = the entry points, by _glapi_get_proc_address), so that GLReference drives either."""

    def __init__(self):
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        lib = os.path.join(here, '_build', 'libmesa_headless.so')
        src = os.path.join(here, 'mesa_headless.c')
        if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
            os.makedirs(os.path.dirname(lib), exist_ok=True)
            tmp = '%s.%d.tmp' % (lib, os.getpid())
            subprocess.check_call(['gcc', '-shared', '-fPIC', '-O1', '-o', tmp, src, '-ldl'])
            os.replace(tmp, lib)
        self.lib = ctypes.CDLL(lib)
        self.lib.mesa_headless_error.restype = ctypes.c_char_p
        self.lib.mesa_headless_proc.restype = ctypes.c_void_p
        self.lib.mesa_headless_proc.argtypes = [ctypes.c_char_p]
        if self.lib.mesa_headless_init(MESA_DRIVER.encode(), MESA_GLAPI.encode(), 3, 3) != 0:
            raise RuntimeError('mesa_headless_init: ' + self.lib.mesa_headless_error().decode())
        vp, ci, cu, cf, cb = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_float, ctypes.c_ubyte
        pu, pi, cs = ctypes.POINTER(cu), ctypes.POINTER(ci), ctypes.c_char_p
        table = {
            'glGetString': (cs, [cu]), 'glGetError': (cu, []), 'glGetIntegerv': (None, [cu, pi]),
            'glCreateShader': (cu, [cu]), 'glShaderSource': (None, [cu, ci, ctypes.POINTER(cs), vp]), 'glCompileShader': (None, [cu]),
            'glGetShaderiv': (None, [cu, cu, pi]), 'glGetProgramiv': (None, [cu, cu, pi]),
            'glGetShaderInfoLog': (None, [cu, ci, vp, cs]), 'glGetProgramInfoLog': (None, [cu, ci, vp, cs]),
            'glCreateProgram': (cu, []), 'glAttachShader': (None, [cu, cu]), 'glLinkProgram': (None, [cu]), 'glUseProgram': (None, [cu]),
            'glGetUniformLocation': (ci, [cu, cs]), 'glGetAttribLocation': (ci, [cu, cs]),
            'glGenBuffers': (None, [ci, pu]), 'glGenTextures': (None, [ci, pu]), 'glGenFramebuffers': (None, [ci, pu]),
            'glGenRenderbuffers': (None, [ci, pu]), 'glGenVertexArrays': (None, [ci, pu]),
            'glDeleteBuffers': (None, [ci, pu]), 'glDeleteTextures': (None, [ci, pu]), 'glDeleteFramebuffers': (None, [ci, pu]),
            'glDeleteRenderbuffers': (None, [ci, pu]),
            'glBindBuffer': (None, [cu, cu]), 'glBufferData': (None, [cu, ctypes.c_ssize_t, vp, cu]), 'glBindVertexArray': (None, [cu]),
            'glBindTexture': (None, [cu, cu]), 'glActiveTexture': (None, [cu]), 'glPixelStorei': (None, [cu, ci]),
            'glTexImage2D': (None, [cu, ci, ci, ci, ci, ci, cu, cu, vp]), 'glTexParameteri': (None, [cu, cu, ci]),
            'glTexBuffer': (None, [cu, cu, cu]),
            'glBindFramebuffer': (None, [cu, cu]), 'glBindRenderbuffer': (None, [cu, cu]),
            'glRenderbufferStorage': (None, [cu, cu, ci, ci]), 'glFramebufferRenderbuffer': (None, [cu, cu, cu, cu]),
            'glCheckFramebufferStatus': (cu, [cu]),
            'glEnableVertexAttribArray': (None, [cu]), 'glDisableVertexAttribArray': (None, [cu]),
            'glVertexAttribPointer': (None, [cu, ci, cu, cb, ci, vp]), 'glVertexAttribIPointer': (None, [cu, ci, cu, ci, vp]),
            'glUniformMatrix4fv': (None, [ci, ci, cb, vp]), 'glUniform1f': (None, [ci, cf]), 'glUniform2f': (None, [ci, cf, cf]),
            'glUniform1i': (None, [ci, ci]),
            'glViewport': (None, [ci, ci, ci, ci]), 'glEnable': (None, [cu]), 'glDisable': (None, [cu]), 'glDepthFunc': (None, [cu]),
            'glDepthMask': (None, [cb]), 'glFrontFace': (None, [cu]), 'glCullFace': (None, [cu]),
            'glClearColor': (None, [cf, cf, cf, cf]), 'glClearDepthf': (None, [cf]), 'glClear': (None, [cu]),
            'glDrawElements': (None, [cu, ci, cu, vp]), 'glDrawArrays': (None, [cu, ci, ci]),
            'glReadPixels': (None, [ci, ci, ci, ci, cu, cu, vp]), 'glFinish': (None, []),
        }
        g = self.gles = _Namespace()
        for name, (res, args) in table.items():
            addr = self.lib.mesa_headless_proc(name.encode())
            if not addr:
                raise RuntimeError('Mesa does not export ' + name)
            setattr(g, name, ctypes.CFUNCTYPE(res, *args)(addr))
        self.version = g.glGetString(0x1F02).decode()
        self.renderer = g.glGetString(0x1F01).decode()
        self.glsl_version = g.glGetString(0x8B8C).decode()
        bits = ci()
        g.glGetIntegerv(0x0D50, ctypes.byref(bits))
        self.subpixel_bits = bits.value
        vao = cu()
        g.glGenVertexArrays(1, ctypes.byref(vao))   # the core profile has no default vertex array object
        g.glBindVertexArray(vao.value)

    gen = _GL.gen
    check = _GL.check


_backends = {}
_active = 'swiftshader'


def use_backend(name):
    """makes `name` ('swiftshader' | 'mesa') the GL that gl() returns (each has its own context, created on first use)"""
    global _active
    assert name in ('swiftshader', 'mesa')
    _active = name


def gl():
    if _active not in _backends:
        _backends[_active] = _GLMesa() if _active == 'mesa' else _GL()
    return _backends[_active]


def _program(vert_src, frag_src):
    g = gl().gles
    ids = []
    for kind, src in ((GL_VERTEX_SHADER, vert_src), (GL_FRAGMENT_SHADER, frag_src)):
        s = g.glCreateShader(kind)
        b = ctypes.c_char_p(src.encode())
        g.glShaderSource(s, 1, ctypes.byref(b), None)
        g.glCompileShader(s)
        ok = ctypes.c_int()
        g.glGetShaderiv(s, GL_COMPILE_STATUS, ctypes.byref(ok))
        if not ok.value:
            log = ctypes.create_string_buffer(4096)
            g.glGetShaderInfoLog(s, 4096, None, log)
            raise RuntimeError('shader compile failed: %s\n%s' % (log.value.decode(), src))
        ids.append(s)
    p = g.glCreateProgram()
    for s in ids:
        g.glAttachShader(p, s)
    g.glLinkProgram(p)
    ok = ctypes.c_int()
    g.glGetProgramiv(p, GL_LINK_STATUS, ctypes.byref(ok))
    if not ok.value:
        log = ctypes.create_string_buffer(4096)
        g.glGetProgramInfoLog(p, 4096, None, log)
        raise RuntimeError('program link failed: %s' % log.value.decode())
    return p


def _texture(data, w, h, internal, fmt, wrap):
    G = gl()
    g = G.gles
    t = G.gen(g.glGenTextures)
    g.glBindTexture(GL_TEXTURE_2D, t)
    g.glPixelStorei(GL_UNPACK_ALIGNMENT, 1)
    data = np.ascontiguousarray(data)
    g.glTexImage2D(GL_TEXTURE_2D, 0, internal, w, h, 0, fmt, GL_UNSIGNED_BYTE, data.ctypes.data_as(ctypes.c_void_p))
    for pname, val in ((GL_TEXTURE_MIN_FILTER, GL_NEAREST), (GL_TEXTURE_MAG_FILTER, GL_NEAREST),
                       (GL_TEXTURE_WRAP_S, wrap), (GL_TEXTURE_WRAP_T, wrap)):
        g.glTexParameteri(GL_TEXTURE_2D, pname, val)
    G.check('texture upload')
    return t


def _buffer(target, data):
    G = gl()
    g = G.gles
    b = G.gen(g.glGenBuffers)
    g.glBindBuffer(target, b)
    data = np.ascontiguousarray(data)
    g.glBufferData(target, data.nbytes, data.ctypes.data_as(ctypes.c_void_p), GL_STATIC_DRAW)
    return b


def _rg8(a16):
    """u16 texels (lo = palette index, hi = 0xFF transparent) as the byte pairs ClientFormat::U8U8 uploads."""
    return np.ascontiguousarray(a16, '<u2').view(np.uint8)


class GLReference:
    """The reference's GL draw path for one level (arrays keyed like BuiltLevel.arrays())."""

    def __init__(self, lvl, backend='swiftshader'):
        get = (lambda k, d=None: lvl.get(k, d)) if isinstance(lvl, dict) else (lambda k, d=None: getattr(lvl, k, d))
        self.backend = backend
        use_backend(backend)
        G = gl()
        g = G.gles
        self.draws = np.asarray(get('draws'), np.uint32).reshape(-1, 4)
        sv = np.ascontiguousarray(get('static_vertices'))
        dv = np.ascontiguousarray(get('decor_vertices', np.zeros(0, np.uint8)))
        kv = np.ascontiguousarray(get('sky_vertices'), np.float32).reshape(-1, 3)
        si = np.ascontiguousarray(get('static_indices'), np.uint32)
        di = np.ascontiguousarray(get('decor_indices', np.zeros(0, np.uint32)), np.uint32)
        ki = np.ascontiguousarray(get('sky_indices'), np.uint32)
        assert sv.dtype.itemsize == 48 and (dv.size == 0 or dv.dtype.itemsize == 44)
        # primitive numbering = the order of the draw list (== the HIP renderer's and the oracle's primitive ids)
        self.first_prim = np.zeros(len(self.draws), np.int64)
        n = 0
        for i, (_k, _o, _f, count) in enumerate(self.draws):
            self.first_prim[i] = n
            n += int(count) // 3
        self.n_prims = n
        self.vbuf = {'static': _buffer(GL_ARRAY_BUFFER, sv), 'sprite': _buffer(GL_ARRAY_BUFFER, dv if dv.size else np.zeros(44, np.uint8)),
                     'sky': _buffer(GL_ARRAY_BUFFER, kv if kv.size else np.zeros(3, np.float32))}
        self.ibuf = {'static': _buffer(GL_ELEMENT_ARRAY_BUFFER, si if si.size else np.zeros(3, np.uint32)),
                     'sprite': _buffer(GL_ELEMENT_ARRAY_BUFFER, di if di.size else np.zeros(3, np.uint32)),
                     'sky': _buffer(GL_ELEMENT_ARRAY_BUFFER, ki if ki.size else np.zeros(3, np.uint32))}
        # de-indexed copies + per-primitive id colours for the auxiliary pass
        self.id_vbuf, self.id_cbuf, self.id_first = {}, {}, {}
        for name, verts, idx, kinds in (('static', sv, si, (KIND_FLAT, KIND_WALL)), ('sprite', dv, di, (KIND_DECOR,)),
                                        ('sky', kv, ki, (KIND_SKY,))):
            if idx.size == 0:
                continue
            flat = np.ascontiguousarray(verts[idx])
            ids = np.zeros((len(idx), 3), np.float32)
            for i, (kind, _o, first, count) in enumerate(self.draws):
                if int(kind) in kinds:
                    pid = self.first_prim[i] + np.arange(int(count) // 3)
                    rgb = np.stack([pid & 255, (pid >> 8) & 255, (pid >> 16) & 255], 1).astype(np.float32) / np.float32(255.0)
                    ids[int(first):int(first) + int(count)] = np.repeat(rgb, 3, axis=0)
            self.id_vbuf[name] = _buffer(GL_ARRAY_BUFFER, flat)
            self.id_cbuf[name] = _buffer(GL_ARRAY_BUFFER, ids)
        fa, wa, da, st = (np.asarray(get(k)) for k in ('flat_atlas', 'wall_atlas', 'decor_atlas', 'sky_texture'))
        self.tex = {}
        self.size = {}
        if fa.size:
            self.tex['flat'] = _texture(fa.astype(np.uint8), fa.shape[1], fa.shape[0], GL_R8, GL_RED, GL_REPEAT)
            self.size['flat'] = (fa.shape[1], fa.shape[0])
        for key, a in (('wall', wa), ('decor', da), ('sky', st)):
            if a.size:
                self.tex[key] = _texture(_rg8(a), a.shape[1], a.shape[0], GL_RG8, GL_RG, GL_REPEAT)
                self.size[key] = (a.shape[1], a.shape[0])
        pal = np.asarray(get('palette'), np.uint8).reshape(256, 3)
        cmap = np.asarray(get('colormap'), np.uint8).reshape(32, 256)
        self.palette_rgb = pal[cmap]   # build_palette_texture(0, 0, 32): rgb[row][i] = PLAYPAL0[COLORMAP[row][i]]
        self.playpal = pal
        self.tex['palette'] = _texture(self.palette_rgb, 256, 32, GL_RGB8, GL_RGB, GL_CLAMP_TO_EDGE)
        self.lights_tex = G.gen(g.glGenTextures)
        self.sky_band = float(get('sky_band', 0.0))
        self.programs, self.sources = {}, {}
        if backend == 'mesa':   # u_lights is the reference's samplerBuffer: a GL_R8 buffer texture (game_shaders.rs:152-159)
            self.lights_buf = G.gen(g.glGenBuffers)
        for mode in ('colour', 'ids', 'varyings'):
            for name in ('static', 'sky', 'sprite'):
                src = {}
                for stage in ('vert', 'frag'):
                    with open(os.path.join(REFERENCE_SHADERS, '%s.%s' % (name, stage))) as f:
                        src[stage] = patch_shader(f.read(), stage, mode, backend)
                self.programs[(name, mode)] = _program(src['vert'], src['frag'])
                self.sources[(name, mode)] = src
        self.fbos = {}
        G.check('level upload')

    def _target(self, w, h, colour_format=GL_RGBA8):
        G = gl()
        g = G.gles
        if (w, h, colour_format) in self.fbos:
            g.glBindFramebuffer(GL_FRAMEBUFFER, self.fbos[(w, h, colour_format)])
            return
        fbo = self.fbos[(w, h, colour_format)] = G.gen(g.glGenFramebuffers)
        g.glBindFramebuffer(GL_FRAMEBUFFER, fbo)
        for attach, fmt in ((GL_COLOR_ATTACHMENT0, colour_format), (GL_DEPTH_ATTACHMENT, GL_DEPTH_COMPONENT24)):
            rb = G.gen(g.glGenRenderbuffers)
            g.glBindRenderbuffer(GL_RENDERBUFFER, rb)
            g.glRenderbufferStorage(GL_RENDERBUFFER, fmt, w, h)
            g.glFramebufferRenderbuffer(GL_FRAMEBUFFER, attach, GL_RENDERBUFFER, rb)
        if g.glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE:
            raise RuntimeError('framebuffer incomplete')

    def _attribs(self, prog, layout, stride, ids_name=None):
        g = gl().gles
        for name, comps, typ, offset, integer in layout:
            loc = g.glGetAttribLocation(prog, name.encode())
            if loc < 0:
                continue
            g.glEnableVertexAttribArray(loc)
            if integer:
                g.glVertexAttribIPointer(loc, comps, typ, stride, ctypes.c_void_p(offset))
            else:
                g.glVertexAttribPointer(loc, comps, typ, 0, stride, ctypes.c_void_p(offset))

    STATIC_LAYOUT = [('a_pos', 3, GL_FLOAT, 0, False), ('a_atlas_uv', 2, GL_FLOAT, 12, False),
                     ('a_tile_uv', 2, GL_FLOAT, 20, False), ('a_tile_size', 2, GL_FLOAT, 28, False),
                     ('a_scroll_rate', 1, GL_FLOAT, 36, False), ('a_row_height', 1, GL_FLOAT, 40, False),
                     ('a_num_frames', 1, GL_UNSIGNED_BYTE, 44, True), ('a_light', 1, GL_UNSIGNED_BYTE, 45, True)]
    SPRITE_LAYOUT = [('a_pos', 3, GL_FLOAT, 0, False), ('a_atlas_uv', 2, GL_FLOAT, 12, False),
                     ('a_tile_uv', 2, GL_FLOAT, 20, False), ('a_tile_size', 2, GL_FLOAT, 28, False),
                     ('a_local_x', 1, GL_FLOAT, 36, False), ('a_num_frames', 1, GL_UNSIGNED_BYTE, 40, True),
                     ('a_light', 1, GL_UNSIGNED_BYTE, 41, True)]
    SKY_LAYOUT = [('a_pos', 3, GL_FLOAT, 0, False)]

    def render(self, modelview, projection, time, lights, width, height, mode='colour', object_modelviews=None,
               kinds=0xF):
        """One frame exactly as Renderer::update draws it.  Returns (height, width, 3) u8, row 0 = bottom (glReadPixels).
        mode 'ids': auxiliary pass, returns (height, width) primitive ids, NO_PRIM where nothing was drawn.
        mode 'varyings': auxiliary pass into an RGBA32F target, returns (height, width, 3) float32 =
        (v_tile_uv.x, v_tile_uv.y, v_dist) for static / sprite fragments, (folded sky uv, -1) for sky fragments."""
        use_backend(self.backend)
        G = gl()
        g = G.gles
        ids = mode == 'ids'
        self._target(width, height, GL_RGBA32F if mode == 'varyings' else GL_RGBA8)
        g.glViewport(0, 0, width, height)
        g.glDisable(GL_DITHER)
        g.glDisable(GL_BLEND)
        g.glEnable(GL_DEPTH_TEST)
        g.glDepthFunc(GL_LESS)
        g.glDepthMask(1)
        g.glEnable(GL_CULL_FACE)      # CullClockwise: counter-clockwise is front, back faces culled
        g.glFrontFace(GL_CCW)
        g.glCullFace(GL_BACK)
        if mode != 'colour':
            g.glClearColor(1.0, 1.0, 1.0, 1.0)
        else:
            g.glClearColor(0.06, 0.07, 0.09, 0.0)
        g.glClearDepthf(1.0)
        g.glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT)
        # u_lights: 256 normalised u8
        g.glActiveTexture(GL_TEXTURE0 + 3)
        li = np.ascontiguousarray(lights, np.uint8).reshape(256)
        if self.backend == 'mesa':
            g.glBindBuffer(GL_TEXTURE_BUFFER, self.lights_buf)
            g.glBufferData(GL_TEXTURE_BUFFER, 256, li.ctypes.data_as(ctypes.c_void_p), GL_STATIC_DRAW)
            g.glBindTexture(GL_TEXTURE_BUFFER, self.lights_tex)
            g.glTexBuffer(GL_TEXTURE_BUFFER, GL_R8, self.lights_buf)
        else:
            g.glBindTexture(GL_TEXTURE_2D, self.lights_tex)
            g.glPixelStorei(GL_UNPACK_ALIGNMENT, 1)
            g.glTexImage2D(GL_TEXTURE_2D, 0, GL_R8, 256, 1, 0, GL_RED, GL_UNSIGNED_BYTE, li.ctypes.data_as(ctypes.c_void_p))
            for pname, val in ((GL_TEXTURE_MIN_FILTER, GL_NEAREST), (GL_TEXTURE_MAG_FILTER, GL_NEAREST),
                               (GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE), (GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE)):
                g.glTexParameteri(GL_TEXTURE_2D, pname, val)
        pr = np.ascontiguousarray(projection, np.float32).reshape(16)
        mv0 = np.ascontiguousarray(modelview, np.float32).reshape(16)
        om = None if object_modelviews is None else np.ascontiguousarray(object_modelviews, np.float32).reshape(-1, 16)
        for i, (kind, obj, first, count) in enumerate(self.draws):
            kind, first, count = int(kind), int(first), int(count)
            if count == 0 or not (kinds >> kind) & 1:
                continue
            name = {KIND_FLAT: 'static', KIND_WALL: 'static', KIND_DECOR: 'sprite', KIND_SKY: 'sky'}[kind]
            prog = self.programs[(name, mode)]
            g.glUseProgram(prog)
            layout, stride = {'static': (self.STATIC_LAYOUT, 48), 'sprite': (self.SPRITE_LAYOUT, 44),
                              'sky': (self.SKY_LAYOUT, 12)}[name]
            for loc in range(12):
                g.glDisableVertexAttribArray(loc)
            if ids:
                g.glBindBuffer(GL_ARRAY_BUFFER, self.id_vbuf[name])
                self._attribs(prog, layout, stride)
                g.glBindBuffer(GL_ARRAY_BUFFER, self.id_cbuf[name])
                loc = g.glGetAttribLocation(prog, b'x_id')
                g.glEnableVertexAttribArray(loc)
                g.glVertexAttribPointer(loc, 3, GL_FLOAT, 0, 12, ctypes.c_void_p(0))
                g.glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, 0)
            else:
                g.glBindBuffer(GL_ARRAY_BUFFER, self.vbuf[name])
                self._attribs(prog, layout, stride)
                g.glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, self.ibuf[name])
            mv = mv0 if om is None else om[int(obj)]
            u = lambda n: g.glGetUniformLocation(prog, n)  # noqa: E731
            g.glUniformMatrix4fv(u(b'u_modelview'), 1, 0, mv.ctypes.data_as(ctypes.c_void_p))
            g.glUniformMatrix4fv(u(b'u_projection'), 1, 0, pr.ctypes.data_as(ctypes.c_void_p))
            g.glUniform1f(u(b'u_time'), float(time))
            atlas = {KIND_FLAT: 'flat', KIND_WALL: 'wall', KIND_DECOR: 'decor', KIND_SKY: 'sky'}[kind]
            g.glActiveTexture(GL_TEXTURE0)
            g.glBindTexture(GL_TEXTURE_2D, self.tex[atlas])
            g.glActiveTexture(GL_TEXTURE0 + 1)
            g.glBindTexture(GL_TEXTURE_2D, self.tex['palette'])
            if kind == KIND_SKY:
                g.glUniform1i(u(b'u_texture'), 0)
                g.glUniform1f(u(b'u_tiled_band_size'), self.sky_band)
            else:
                g.glUniform1i(u(b'u_atlas'), 0)
                g.glUniform2f(u(b'u_atlas_size'), float(self.size[atlas][0]), float(self.size[atlas][1]))
                g.glUniform1i(u(b'u_lights'), 3)
            g.glUniform1i(u(b'u_palette'), 1)
            if ids:
                g.glDrawArrays(GL_TRIANGLES, first, count)
            else:
                g.glDrawElements(GL_TRIANGLES, count, GL_UNSIGNED_INT, ctypes.c_void_p(4 * first))
        g.glPixelStorei(GL_PACK_ALIGNMENT, 1)
        if mode == 'varyings':
            var = np.zeros((height, width, 4), np.float32)
            g.glReadPixels(0, 0, width, height, GL_RGBA, GL_FLOAT, var.ctypes.data_as(ctypes.c_void_p))
            G.check('render varyings')
            return var[..., :3].copy()
        out = np.zeros((height, width, 4), np.uint8)
        g.glReadPixels(0, 0, width, height, GL_RGBA, GL_UNSIGNED_BYTE, out.ctypes.data_as(ctypes.c_void_p))
        G.check('render')
        if ids:
            return out[..., 0].astype(np.uint32) | (out[..., 1].astype(np.uint32) << 8) | (out[..., 2].astype(np.uint32) << 16)
        return out[..., :3].copy()
