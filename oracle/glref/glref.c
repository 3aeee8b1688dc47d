/*
 * glref — headless llvmpipe runner for the reference's OWN fragment shaders.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md). Nothing in the product path
 * may link, load or execute this file. It exists so that the reference GLSL
 * (src/ssgi/shader/ssgi.frag, src/temporal-reproject/shader/temporal_reproject.frag,
 * src/denoise/shader/poisson_denoise.frag, DenoiserComposePass.js inline shader)
 * can be executed UNMODIFIED on the CPU by Mesa llvmpipe and its outputs used as
 * golden vectors / as the "reference" CPU baseline.
 *
 * No X server, EGL or OSMesa is needed: the DRI swrast driver is loaded directly
 * through its DRI_SWRast loader interface and GL entry
 * points are resolved with _glapi_get_proc_address.
 *
 * The library is a thin, generic "compile program / make texture / set uniform /
 * draw full-screen triangle into FBO / read back" layer; the pass orchestration
 * (which uniforms, which textures, ping-pong) is done by oracle/glref/chain.py,
 * which mirrors the reference's JS drivers.
 *
 * Environment required by the caller (set before the first call, chain.py does it):
 *   glsl_zero_init=true   WebGL zero-initialises locals; raw Mesa does not
 *                         (ssgi.frag:140-151 relies on it)
 *   LP_NUM_THREADS=<n>    llvmpipe rasteriser threads
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <dlfcn.h>
#include <time.h>
#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>

static void getDrawableInfo(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p) { (void)d; (void)p; *x = 0; *y = 0; *w = 64; *h = 64; }
static void putImage(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void getImage(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p) { (void)d; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void putImage2(__DRIdrawable *d, int op, int x, int y, int w, int h, int stride, char *data, void *p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }
static void getImage2(__DRIdrawable *d, int x, int y, int w, int h, int stride, char *data, void *p) { (void)d; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }

static const __DRIswrastLoaderExtension swrastLoader = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = getDrawableInfo,
    .putImage = putImage,
    .getImage = getImage,
    .putImage2 = putImage2,
    .getImage2 = getImage2,
};
static const __DRIextension *loader_exts[] = {&swrastLoader.base, NULL};

typedef void *(*getproc_t)(const char *);
static getproc_t gp;

#define DECL(T, n) static T p##n
DECL(PFNGLCREATESHADERPROC, CreateShader);
DECL(PFNGLSHADERSOURCEPROC, ShaderSource);
DECL(PFNGLCOMPILESHADERPROC, CompileShader);
DECL(PFNGLGETSHADERIVPROC, GetShaderiv);
DECL(PFNGLGETSHADERINFOLOGPROC, GetShaderInfoLog);
DECL(PFNGLCREATEPROGRAMPROC, CreateProgram);
DECL(PFNGLATTACHSHADERPROC, AttachShader);
DECL(PFNGLLINKPROGRAMPROC, LinkProgram);
DECL(PFNGLUSEPROGRAMPROC, UseProgram);
DECL(PFNGLGETPROGRAMIVPROC, GetProgramiv);
DECL(PFNGLGETPROGRAMINFOLOGPROC, GetProgramInfoLog);
DECL(PFNGLBINDATTRIBLOCATIONPROC, BindAttribLocation);
DECL(PFNGLGENFRAMEBUFFERSPROC, GenFramebuffers);
DECL(PFNGLBINDFRAMEBUFFERPROC, BindFramebuffer);
DECL(PFNGLFRAMEBUFFERTEXTURE2DPROC, FramebufferTexture2D);
DECL(PFNGLCHECKFRAMEBUFFERSTATUSPROC, CheckFramebufferStatus);
DECL(PFNGLGENVERTEXARRAYSPROC, GenVertexArrays);
DECL(PFNGLBINDVERTEXARRAYPROC, BindVertexArray);
DECL(PFNGLGENBUFFERSPROC, GenBuffers);
DECL(PFNGLBINDBUFFERPROC, BindBuffer);
DECL(PFNGLBUFFERDATAPROC, BufferData);
DECL(PFNGLVERTEXATTRIBPOINTERPROC, VertexAttribPointer);
DECL(PFNGLENABLEVERTEXATTRIBARRAYPROC, EnableVertexAttribArray);
DECL(PFNGLDRAWBUFFERSPROC, DrawBuffers);
DECL(PFNGLGETUNIFORMLOCATIONPROC, GetUniformLocation);
DECL(PFNGLUNIFORM1IPROC, Uniform1i);
DECL(PFNGLUNIFORM1FPROC, Uniform1f);
DECL(PFNGLUNIFORM2FPROC, Uniform2f);
DECL(PFNGLUNIFORM3FPROC, Uniform3f);
DECL(PFNGLUNIFORMMATRIX4FVPROC, UniformMatrix4fv);
DECL(PFNGLACTIVETEXTUREPROC, ActiveTexture);
DECL(PFNGLGETACTIVEUNIFORMPROC, GetActiveUniform);
DECL(PFNGLGENERATEMIPMAPPROC, GenerateMipmap);
static void (*pGenTextures)(GLsizei, GLuint *);
static void (*pBindTexture)(GLenum, GLuint);
static void (*pTexImage2D)(GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *);
static void (*pTexParameteri)(GLenum, GLenum, GLint);
static void (*pViewport)(GLint, GLint, GLsizei, GLsizei);
static void (*pDrawArrays)(GLenum, GLint, GLsizei);
static void (*pFinish)(void);
static GLenum (*pGetError)(void);
static const GLubyte *(*pGetString)(GLenum);
static void (*pReadPixels)(GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, void *);
static void (*pReadBuffer)(GLenum);
static void (*pDisable)(GLenum);
static void (*pPixelStorei)(GLenum, GLint);
static void (*pDeleteTextures)(GLsizei, const GLuint *);

#define MAX_TEX 16384
#define MAX_PROG 2048
#define MAX_SAMP 24

typedef struct {
    GLuint id;
    int w, h, fmt;
    int cube; /* 1: GL_TEXTURE_CUBE_MAP (sampler binding only; never a render target) */
} tex_t;
typedef struct {
    GLuint id;
    int nsamp;
    GLint samp_loc[MAX_SAMP];
    int samp_tex[MAX_SAMP];
    char samp_name[MAX_SAMP][64];
} prog_t;

static tex_t g_tex[MAX_TEX];
static int g_ntex = 1; /* 0 = invalid */
static prog_t g_prog[MAX_PROG];
static int g_nprog = 1;
static GLuint g_fbo, g_rfbo, g_vao, g_vbo;
static int g_init = 0;
static char g_info[512];
static double g_last_ms = 0;

enum { FMT_RGBA32F = 0, FMT_RGBA16F = 1, FMT_R32F = 2, FMT_RGBA8 = 3 };

int glref_init(void) {
    if (g_init) return 0;
    void *glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    const char *drvpath = getenv("GLREF_SWRAST");
    if (!drvpath) drvpath = "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so";
    void *drv = dlopen(drvpath, RTLD_NOW | RTLD_GLOBAL);
    if (!glapi || !drv) { snprintf(g_info, sizeof g_info, "dlopen failed: %s", dlerror()); return -1; }
    const __DRIextension **(*getExt)(void) = dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!getExt) { snprintf(g_info, sizeof g_info, "no __driDriverGetExtensions_swrast"); return -2; }
    const __DRIextension **exts = getExt();
    const __DRIcoreExtension *core = NULL;
    const __DRIswrastExtension *sw = NULL;
    for (int i = 0; exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const void *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) sw = (const void *)exts[i];
    }
    if (!core || !sw) { snprintf(g_info, sizeof g_info, "DRI_Core/DRI_SWRast missing"); return -3; }
    const __DRIconfig **configs = NULL;
    __DRIscreen *scr = sw->createNewScreen2(0, loader_exts, exts, &configs, NULL);
    if (!scr) { snprintf(g_info, sizeof g_info, "createNewScreen2 failed"); return -4; }
    uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 3, __DRI_CTX_ATTRIB_MINOR_VERSION, 0};
    unsigned err = 0;
    __DRIcontext *ctx = sw->createContextAttribs(scr, __DRI_API_GLES3, configs[0], NULL, 2, attribs, &err, NULL);
    if (!ctx) { snprintf(g_info, sizeof g_info, "createContextAttribs failed err=%u", err); return -5; }
    __DRIdrawable *dr = sw->createNewDrawable(scr, configs[0], NULL);
    if (!core->bindContext(ctx, dr, dr)) { snprintf(g_info, sizeof g_info, "bindContext failed"); return -6; }
    gp = (getproc_t)dlsym(glapi, "_glapi_get_proc_address");
#define L(n) p##n = (void *)gp("gl" #n)
    L(CreateShader); L(ShaderSource); L(CompileShader); L(GetShaderiv); L(GetShaderInfoLog);
    L(CreateProgram); L(AttachShader); L(LinkProgram); L(UseProgram); L(GetProgramiv); L(GetProgramInfoLog);
    L(BindAttribLocation); L(GenFramebuffers); L(BindFramebuffer); L(FramebufferTexture2D); L(CheckFramebufferStatus);
    L(GenVertexArrays); L(BindVertexArray); L(GenBuffers); L(BindBuffer); L(BufferData);
    L(VertexAttribPointer); L(EnableVertexAttribArray); L(DrawBuffers); L(GetUniformLocation);
    L(Uniform1i); L(Uniform1f); L(Uniform2f); L(Uniform3f); L(UniformMatrix4fv); L(ActiveTexture); L(GetActiveUniform); L(GenerateMipmap);
    L(GenTextures); L(BindTexture); L(TexImage2D); L(TexParameteri); L(Viewport); L(DrawArrays); L(Finish);
    L(GetError); L(GetString); L(ReadPixels); L(ReadBuffer); L(Disable); L(PixelStorei); L(DeleteTextures);
    snprintf(g_info, sizeof g_info, "%s | %s | GLSL %s", (const char *)pGetString(GL_VERSION),
             (const char *)pGetString(GL_RENDERER), (const char *)pGetString(GL_SHADING_LANGUAGE_VERSION));
    pGenFramebuffers(1, &g_fbo);
    pGenFramebuffers(1, &g_rfbo);
    pGenVertexArrays(1, &g_vao);
    pBindVertexArray(g_vao);
    /* postprocessing's Pass draws ONE full-screen triangle */
    static const float tri[9] = {-1, -1, 0, 3, -1, 0, -1, 3, 0};
    pGenBuffers(1, &g_vbo);
    pBindBuffer(GL_ARRAY_BUFFER, g_vbo);
    pBufferData(GL_ARRAY_BUFFER, sizeof tri, tri, GL_STATIC_DRAW);
    pVertexAttribPointer(0, 3, GL_FLOAT, GL_FALSE, 0, 0);
    pEnableVertexAttribArray(0);
    pDisable(GL_DEPTH_TEST);
    pDisable(GL_BLEND);
    pDisable(GL_DITHER);
    pPixelStorei(GL_PACK_ALIGNMENT, 1);
    pPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    g_init = 1;
    return 0;
}

const char *glref_info(void) { return g_info; }
double glref_last_ms(void) { return g_last_ms; }

/* vertex stage = src/utils/shader/basic.vert semantics with three's attribute decl */
static const char *VS =
    "#version 300 es\n"
    "precision highp float;\n"
    "in vec3 position;\n"
    "out vec2 vUv;\n"
    "void main() {\n"
    "    vUv = position.xy * 0.5 + 0.5;\n"
    "    gl_Position = vec4(position.xy, 1.0, 1.0);\n"
    "}\n";

int glref_program(const char *fs_src, char *log, int loglen) {
    if (g_nprog >= MAX_PROG) return -1;
    GLint ok;
    GLuint v = pCreateShader(GL_VERTEX_SHADER), f = pCreateShader(GL_FRAGMENT_SHADER);
    pShaderSource(v, 1, &VS, NULL);
    pCompileShader(v);
    pGetShaderiv(v, GL_COMPILE_STATUS, &ok);
    if (!ok) { if (log) pGetShaderInfoLog(v, loglen, NULL, log); return -2; }
    pShaderSource(f, 1, &fs_src, NULL);
    pCompileShader(f);
    pGetShaderiv(f, GL_COMPILE_STATUS, &ok);
    if (!ok) { if (log) pGetShaderInfoLog(f, loglen, NULL, log); return -3; }
    GLuint p = pCreateProgram();
    pAttachShader(p, v);
    pAttachShader(p, f);
    pBindAttribLocation(p, 0, "position");
    pLinkProgram(p);
    pGetProgramiv(p, GL_LINK_STATUS, &ok);
    if (!ok) { if (log) pGetProgramInfoLog(p, loglen, NULL, log); return -4; }
    prog_t *P = &g_prog[g_nprog];
    memset(P, 0, sizeof *P);
    P->id = p;
    return g_nprog++;
}

/* list active uniforms (debug / completeness check): writes "name:type\n" lines */
int glref_active_uniforms(int prog, char *out, int outlen) {
    GLint nu;
    pGetProgramiv(g_prog[prog].id, GL_ACTIVE_UNIFORMS, &nu);
    int off = 0;
    for (int i = 0; i < nu; i++) {
        char name[256]; GLint sz; GLenum ty;
        pGetActiveUniform(g_prog[prog].id, i, 256, NULL, &sz, &ty, name);
        off += snprintf(out + off, outlen - off > 0 ? outlen - off : 0, "%s:%x\n", name, ty);
    }
    return nu;
}

static void fmt_to_gl(int fmt, GLint *internal, GLenum *format, GLenum *type) {
    switch (fmt) {
    case FMT_RGBA32F: *internal = GL_RGBA32F; *format = GL_RGBA; *type = GL_FLOAT; break;
    case FMT_RGBA16F: *internal = GL_RGBA16F; *format = GL_RGBA; *type = GL_HALF_FLOAT; break;
    case FMT_R32F: *internal = GL_R32F; *format = GL_RED; *type = GL_FLOAT; break;
    default: *internal = GL_RGBA8; *format = GL_RGBA; *type = GL_UNSIGNED_BYTE; break;
    }
}

/* data layout = GL (row 0 = bottom). RGBA16F data is given as raw half bits. NULL -> zeros. */
int glref_texture(int w, int h, int fmt, int linear, int repeat, const void *data) {
    if (g_ntex >= MAX_TEX) return -1;
    GLint internal; GLenum format, type;
    fmt_to_gl(fmt, &internal, &format, &type);
    GLuint t;
    pGenTextures(1, &t);
    pActiveTexture(GL_TEXTURE0 + 31);
    pBindTexture(GL_TEXTURE_2D, t);
    void *zero = NULL;
    if (!data) { zero = calloc((size_t)w * h, 16); data = zero; }
    pTexImage2D(GL_TEXTURE_2D, 0, internal, w, h, 0, format, type, data);
    free(zero);
    pTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, linear ? GL_LINEAR : GL_NEAREST);
    pTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, linear ? GL_LINEAR : GL_NEAREST);
    pTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, repeat ? GL_REPEAT : GL_CLAMP_TO_EDGE);
    pTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, repeat ? GL_REPEAT : GL_CLAMP_TO_EDGE);
    g_tex[g_ntex].id = t; g_tex[g_ntex].w = w; g_tex[g_ntex].h = h; g_tex[g_ntex].fmt = fmt;
    return g_ntex++;
}

/* A cube map for `uniform samplerCube` (CubeToEquirectEnvPass.js:28): six size x size faces in GL order +X -X +Y -Y +Z -Z, back to
 * back in `data`, each row 0 first as handed to glTexImage2D.  filter: 0 NEAREST, 1 LINEAR (no mip chain),
 * 2 LINEAR_MIPMAP_LINEAR + LINEAR with glGenerateMipmap (three's CubeTexture default), 3 the same filters with NO generation: the
 * caller uploads every level (glref_cube_upload_level; probes).  ES 3 contexts filter cube maps seamlessly. */
int glref_cube_texture(int size, int fmt, int filter, const void *data) {
    if (g_ntex >= MAX_TEX) return -1;
    GLint internal; GLenum format, type;
    fmt_to_gl(fmt, &internal, &format, &type);
    const size_t texel = fmt == FMT_RGBA32F ? 16 : (fmt == FMT_RGBA16F ? 8 : (fmt == FMT_R32F ? 4 : 4));
    GLuint t;
    pGenTextures(1, &t);
    pActiveTexture(GL_TEXTURE0 + 31);
    pBindTexture(GL_TEXTURE_CUBE_MAP, t);
    for (int f = 0; f < 6; f++)
        pTexImage2D(GL_TEXTURE_CUBE_MAP_POSITIVE_X + f, 0, internal, size, size, 0, format, type, (const char *)data + (size_t)f * size * size * texel);
    pTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_MIN_FILTER, filter == 0 ? GL_NEAREST : (filter == 1 ? GL_LINEAR : GL_LINEAR_MIPMAP_LINEAR));
    pTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_MAG_FILTER, filter == 0 ? GL_NEAREST : GL_LINEAR);
    pTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    pTexParameteri(GL_TEXTURE_CUBE_MAP, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    if (filter == 2) pGenerateMipmap(GL_TEXTURE_CUBE_MAP);
    g_tex[g_ntex].id = t; g_tex[g_ntex].w = size; g_tex[g_ntex].h = size; g_tex[g_ntex].fmt = fmt; g_tex[g_ntex].cube = 1;
    return pGetError() ? -(int)pGetError() - 100 : g_ntex++;
}

/* One mip level of one face (level size = max(size >> level, 1)). */
int glref_cube_upload_level(int tex, int face, int level, const void *data) {
    tex_t *T = &g_tex[tex];
    GLint internal; GLenum format, type;
    fmt_to_gl(T->fmt, &internal, &format, &type);
    int w = T->w >> level;
    if (w < 1) w = 1;
    pActiveTexture(GL_TEXTURE0 + 31);
    pBindTexture(GL_TEXTURE_CUBE_MAP, T->id);
    pTexImage2D(GL_TEXTURE_CUBE_MAP_POSITIVE_X + face, level, internal, w, w, 0, format, type, data);
    return (int)pGetError();
}

/* Read one face / level back as RGBA float32. */
int glref_cube_read(int tex, int face, int level, float *out) {
    tex_t *T = &g_tex[tex];
    int w = T->w >> level;
    if (w < 1) w = 1;
    pBindFramebuffer(GL_FRAMEBUFFER, g_rfbo);
    pFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_CUBE_MAP_POSITIVE_X + face, T->id, level);
    pReadBuffer(GL_COLOR_ATTACHMENT0);
    GLenum st = pCheckFramebufferStatus(GL_FRAMEBUFFER);
    if (st != GL_FRAMEBUFFER_COMPLETE) return -(int)st;
    pReadPixels(0, 0, w, w, GL_RGBA, GL_FLOAT, out);
    pFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, 0, 0);
    return (int)pGetError();
}

int glref_tex_upload(int tex, const void *data) {
    tex_t *T = &g_tex[tex];
    GLint internal; GLenum format, type;
    fmt_to_gl(T->fmt, &internal, &format, &type);
    pActiveTexture(GL_TEXTURE0 + 31);
    pBindTexture(GL_TEXTURE_2D, T->id);
    pTexImage2D(GL_TEXTURE_2D, 0, internal, T->w, T->h, 0, format, type, data);
    return (int)pGetError();
}

/* three.js `generateMipmaps = true; minFilter = LinearMipMapLinearFilter; magFilter = LinearFilter` (SSGIEffect.js:323-328):
 * the driver builds the chain (glGenerateMipmap), sampling is trilinear. */
int glref_gen_mipmaps(int tex) {
    tex_t *T = &g_tex[tex];
    pActiveTexture(GL_TEXTURE0 + 31);
    pBindTexture(GL_TEXTURE_2D, T->id);
    pTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR_MIPMAP_LINEAR);
    pTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    pGenerateMipmap(GL_TEXTURE_2D);
    return (int)pGetError();
}

/* Overwrite one mip level (probes: a chain whose level l holds the constant l makes textureLod / texture return the lod they used). */
int glref_tex_upload_level(int tex, int level, const void *data) {
    tex_t *T = &g_tex[tex];
    GLint internal; GLenum format, type;
    fmt_to_gl(T->fmt, &internal, &format, &type);
    int w = T->w >> level, h = T->h >> level;
    if (w < 1) w = 1;
    if (h < 1) h = 1;
    pActiveTexture(GL_TEXTURE0 + 31);
    pBindTexture(GL_TEXTURE_2D, T->id);
    pTexImage2D(GL_TEXTURE_2D, level, internal, w, h, 0, format, type, data);
    return (int)pGetError();
}

/* Read mip level `level` of a texture as RGBA float32 (max(w>>level,1) * max(h>>level,1) * 4 floats). */
int glref_read_level(int tex, int level, float *out) {
    tex_t *T = &g_tex[tex];
    int w = T->w >> level, h = T->h >> level;
    if (w < 1) w = 1;
    if (h < 1) h = 1;
    pBindFramebuffer(GL_FRAMEBUFFER, g_rfbo);
    pFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, T->id, level);
    pReadBuffer(GL_COLOR_ATTACHMENT0);
    GLenum st = pCheckFramebufferStatus(GL_FRAMEBUFFER);
    if (st != GL_FRAMEBUFFER_COMPLETE) return -(int)st;
    pReadPixels(0, 0, w, h, GL_RGBA, GL_FLOAT, out);
    pFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, T->id, 0);
    return (int)pGetError();
}

int glref_tex_free(int tex) {
    pDeleteTextures(1, &g_tex[tex].id);
    g_tex[tex].id = 0;
    return 0;
}

int glref_bind_sampler(int prog, const char *name, int tex) {
    prog_t *P = &g_prog[prog];
    for (int i = 0; i < P->nsamp; i++)
        if (!strcmp(P->samp_name[i], name)) { P->samp_tex[i] = tex; return 0; }
    GLint loc = pGetUniformLocation(P->id, name);
    if (loc < 0) return 1; /* not active: harmless */
    if (P->nsamp >= MAX_SAMP) return -1;
    int i = P->nsamp++;
    strncpy(P->samp_name[i], name, 63);
    P->samp_loc[i] = loc;
    P->samp_tex[i] = tex;
    return 0;
}

/* kind: 0=int/bool 1=float 2=vec2 3=vec3 4=mat4 (column-major, as three.js Matrix4.elements) */
int glref_uniform(int prog, const char *name, int kind, const float *v) {
    prog_t *P = &g_prog[prog];
    pUseProgram(P->id);
    GLint loc = pGetUniformLocation(P->id, name);
    if (loc < 0) return 1;
    switch (kind) {
    case 0: pUniform1i(loc, (GLint)v[0]); break;
    case 1: pUniform1f(loc, v[0]); break;
    case 2: pUniform2f(loc, v[0], v[1]); break;
    case 3: pUniform3f(loc, v[0], v[1], v[2]); break;
    case 4: pUniformMatrix4fv(loc, 1, GL_FALSE, v); break;
    default: return -1;
    }
    return 0;
}

int glref_uniform_int(int prog, const char *name, int value) {
    prog_t *P = &g_prog[prog];
    pUseProgram(P->id);
    GLint loc = pGetUniformLocation(P->id, name);
    if (loc < 0) return 1;
    pUniform1i(loc, value);
    return 0;
}

/* Draw the full-screen triangle with `prog` into the given colour attachments.
 * No clear: `discard`ed fragments keep the attachment's previous contents
 * (postprocessing sets renderer.autoClear=false). */
int glref_draw(int prog, const int *targets, int ntargets) {
    prog_t *P = &g_prog[prog];
    pUseProgram(P->id);
    for (int i = 0; i < P->nsamp; i++) {
        pActiveTexture(GL_TEXTURE0 + i);
        pBindTexture(g_tex[P->samp_tex[i]].cube ? GL_TEXTURE_CUBE_MAP : GL_TEXTURE_2D, g_tex[P->samp_tex[i]].id);
        pUniform1i(P->samp_loc[i], i);
    }
    pBindFramebuffer(GL_FRAMEBUFFER, g_fbo);
    GLenum bufs[8];
    for (int k = 0; k < 8; k++) {
        GLuint id = k < ntargets ? g_tex[targets[k]].id : 0;
        pFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0 + k, GL_TEXTURE_2D, id, 0);
        bufs[k] = GL_COLOR_ATTACHMENT0 + k;
    }
    pDrawBuffers(ntargets, bufs);
    GLenum st = pCheckFramebufferStatus(GL_FRAMEBUFFER);
    if (st != GL_FRAMEBUFFER_COMPLETE) return -(int)st;
    pViewport(0, 0, g_tex[targets[0]].w, g_tex[targets[0]].h);
    pBindVertexArray(g_vao);
    struct timespec t0, t1;
    pFinish();
    clock_gettime(CLOCK_MONOTONIC, &t0);
    pDrawArrays(GL_TRIANGLES, 0, 3);
    pFinish();
    clock_gettime(CLOCK_MONOTONIC, &t1);
    g_last_ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) / 1e6;
    return (int)pGetError();
}

/* Read a texture back as RGBA float32 (w*h*4 floats, row 0 = bottom). */
int glref_read(int tex, float *out) {
    tex_t *T = &g_tex[tex];
    pBindFramebuffer(GL_FRAMEBUFFER, g_rfbo);
    pFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, T->id, 0);
    pReadBuffer(GL_COLOR_ATTACHMENT0);
    GLenum st = pCheckFramebufferStatus(GL_FRAMEBUFFER);
    if (st != GL_FRAMEBUFFER_COMPLETE) return -(int)st;
    pReadPixels(0, 0, T->w, T->h, GL_RGBA, GL_FLOAT, out);
    return (int)pGetError();
}
