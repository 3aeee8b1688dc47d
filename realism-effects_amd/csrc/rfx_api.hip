// rfx_api.hip — the C ABI of librfx_hip.so (include/rfx.h): context, texture slots, the four
// draw entry points.  Host side only; kernels live in k1..k4_*.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include "../../include/rfx.h"
#include "rfx_kernels.h"

#include "rfx_ctx.h"

thread_local std::string g_create_err;

// every entry point that takes a context: select its device
#define RFX_ENTER(c) hipSetDevice((c)->device)
// rfx_profile: bracket the launches of one draw with two events on `stream` (no-ops unless profiling is on)
static hipEvent_t prof_event(rfx_ctx *c) {
    hipEvent_t e = nullptr;
    if (!c->prof_free.empty()) { e = c->prof_free.back(); c->prof_free.pop_back(); return e; }
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}
static void prof_recycle(rfx_ctx *c) {
    for (const rfx_ctx::ProfRec &r : c->prof_recs) { c->prof_free.push_back(r.a); c->prof_free.push_back(r.b); }
    c->prof_recs.clear();
}
struct ProfScope {
    rfx_ctx *c;
    hipStream_t stream;
    hipEvent_t a = nullptr, b = nullptr;
    int kind;
    ProfScope(rfx_ctx *c_, int kind_, hipStream_t s) : c(c_), stream(s), kind(kind_) {
        if (!c->profiling || c->prof_recs.size() >= 8192) return;
        a = prof_event(c); b = prof_event(c);
        if (!a || !b || hipEventRecord(a, stream) != hipSuccess) drop();  // (a capturing user stream refuses the record: the draw is simply not timed)
    }
    void drop() { if (a) c->prof_free.push_back(a); if (b) c->prof_free.push_back(b); a = b = nullptr; }
    ~ProfScope() {
        if (!a) return;
        if (hipEventRecord(b, stream) != hipSuccess) { drop(); return; }  // never a half-recorded pair in prof_recs
        c->prof_recs.push_back({kind, a, b});
    }
};

extern "C" {

int rfx_abi_version(void) { return RFX_ABI_VERSION; }

size_t rfx_tex_texel_bytes(rfx_tex id) { return (id >= 0 && id < RFX_TEX_COUNT) ? texel_bytes(id) : 0; }

rfx_ctx *rfx_create(int device, int width, int height, int tile_y0, int tile_rows, int halo_rows) {
    if (width <= 0 || height <= 0 || tile_y0 < 0 || tile_rows <= 0 || tile_y0 + tile_rows > height || halo_rows < 0) {
        fail(nullptr, RFX_EINVAL, "rfx_create: bad geometry");
        return nullptr;
    }
    // kernels address texels with 24-bit multiplies and 32-bit byte offsets: a plane stays < 4 GiB (16K x 16K RGBA32F)
    if (width > 32768 || height > 32768 || (size_t)width * height > ((size_t)1 << 28)) {
        fail(nullptr, RFX_EINVAL, "rfx_create: frames above 32768 in an edge or 2^28 texels are not supported");
        return nullptr;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || device < 0 || device >= ndev) {
        fail(nullptr, RFX_EDEVICE, "rfx_create: no such HIP device", e);
        return nullptr;
    }
    rfx_ctx *c = new rfx_ctx();
    c->device = device;
    c->W = width; c->H = height; c->tile_y0 = tile_y0; c->tile_rows = tile_rows; c->halo = halo_rows;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipMalloc((void **)&c->halo_violations, sizeof(unsigned int)) != hipSuccess) {
        fail(nullptr, RFX_EDEVICE, "rfx_create: stream/event creation failed");
        c->stream = c->own_stream;
        rfx_destroy(c);  // releases whatever was created
        return nullptr;
    }
    hipMemset(c->halo_violations, 0, sizeof(unsigned int));
    c->stream = c->own_stream;
    if (hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->n_cu <= 0) c->n_cu = 256;
    // K1's depth pre-pass stream and the events that order it, created HERE and not lazily by the first draw: every asynchronous writer of
    // the depth slot (rfx_stage_flip, rfx_clear) records ev_depth from the first frame on, so the first pre-pass already waits for the
    // first staged copy (round 3 created them inside the first rfx_ssgi_*: frame 0's pre-pass raced the copy that filled its input)
    if (hipStreamCreateWithFlags(&c->prep_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_depth, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_k1_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_prep_done, hipEventDisableTiming) != hipSuccess) {
        fail(nullptr, RFX_EDEVICE, "rfx_create: K1 pre-pass stream/event creation failed");
        rfx_destroy(c);
        return nullptr;
    }
    const int b0 = tile_y0 - halo_rows < 0 ? 0 : tile_y0 - halo_rows;
    const int b1 = tile_y0 + tile_rows + halo_rows > height ? height : tile_y0 + tile_rows + halo_rows;
    for (int i = 0; i < RFX_TEX_COUNT; i++) {
        Slot &s = c->slots[i];
        s.texel = texel_bytes(i);
        s.width = width;
        // K1 gathers depth and last frame's composed GI anywhere on screen -> held whole
        const bool whole = (i == RFX_TEX_DEPTH || i == RFX_TEX_COMPOSE || i == RFX_TEX_COMPOSE_RGB);
        s.row0 = whole ? 0 : b0;
        s.rows = whole ? height : b1 - b0;
        if (i == RFX_TEX_BLUE_NOISE) { s.row0 = 0; s.rows = 128; s.width = 128; }
    }
    return c;
}

void rfx_destroy(rfx_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    rfx_peer_release(c);
    rfx_comm_release(c);
    // a staged copy may still be writing a back buffer: drain the upload stream before any buffer goes
    if (c->upload_stream) { hipStreamSynchronize(c->upload_stream); hipStreamDestroy(c->upload_stream); }
    for (int i = 0; i < RFX_TEX_COUNT; i++) {
        if (c->slots[i].owned && c->slots[i].ptr) hipFree(c->slots[i].ptr);
        if (c->slots[i].back) hipFree(c->slots[i].back);
    }
    for (hipEvent_t e : c->ev_batch)
        if (e) hipEventDestroy(e);
    if (c->ev_staged) hipEventDestroy(c->ev_staged);
    if (c->ev_frame_done) hipEventDestroy(c->ev_frame_done);
    if (c->halo_violations) hipFree(c->halo_violations);
    if (c->viewz) hipFree(c->viewz);
    if (c->prep_stream) { hipStreamSynchronize(c->prep_stream); hipStreamDestroy(c->prep_stream); }
    for (hipEvent_t e : {c->ev_depth, c->ev_k1_done, c->ev_prep_done})
        if (e) hipEventDestroy(e);
    if (c->hits) hipFree(c->hits);
    if (c->hit_rows_dev) hipFree(c->hit_rows_dev);
    if (c->hit_rows_host) hipHostFree(c->hit_rows_host);
    if (c->hit_mask_dev) hipFree(c->hit_mask_dev);
    if (c->hit_mask_host) hipHostFree(c->hit_mask_host);
    if (c->hist_staging) hipFree(c->hist_staging);
    if (c->coarse) hipFree(c->coarse);
    if (c->cells) hipFree(c->cells);
    if (c->k1_tiles) hipFree(c->k1_tiles);
    if (c->env) hipFree(c->env);
    if (c->env_marginal) hipFree(c->env_marginal);
    if (c->env_conditional) hipFree(c->env_conditional);
    prof_recycle(c);
    for (hipEvent_t e : c->prof_free) hipEventDestroy(e);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
}

const char *rfx_last_error(const rfx_ctx *c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int rfx_get_geometry(const rfx_ctx *c, int *width, int *height, int *tile_y0, int *tile_rows, int *halo_rows) {
    if (!c) return RFX_EINVAL;
    if (width) *width = c->W;
    if (height) *height = c->H;
    if (tile_y0) *tile_y0 = c->tile_y0;
    if (tile_rows) *tile_rows = c->tile_rows;
    if (halo_rows) *halo_rows = c->halo;
    return RFX_OK;
}

int rfx_set_stream(rfx_ctx *c, void *hip_stream) {
    if (!c) return RFX_EINVAL;
    RFX_ENTER(c);
    // work already enqueued (uploads, the zero-fill of fresh render targets) must not race kernels on the new stream
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return RFX_OK;
}

int rfx_set_row_window(rfx_ctx *c, int y0, int y1) {
    if (!c) return RFX_EINVAL;
    RFX_ENTER(c);
    if (y1 <= y0) { c->win_y0 = 0; c->win_y1 = 0x7fffffff; }  // reset
    else { c->win_y0 = y0; c->win_y1 = y1; }
    return RFX_OK;
}

int rfx_set_uv_model(rfx_ctx *c, int model) {
    if (!c) return RFX_EINVAL;
    RFX_ENTER(c);
    if (model != RFX_UV_IDEAL && model != RFX_UV_REFERENCE_GL) return fail(c, RFX_EINVAL, "rfx_set_uv_model: unknown model");
    c->uv_model = model;
    return RFX_OK;
}

int rfx_tex_held_rows(const rfx_ctx *c, rfx_tex id, int *row0, int *rows) {
    if (!c || id < 0 || id >= RFX_TEX_COUNT) return RFX_EINVAL;
    if (row0) *row0 = c->slots[id].row0;
    if (rows) *rows = c->slots[id].rows;
    return RFX_OK;
}

static int ensure(rfx_ctx *c, int id) {
    Slot &s = c->slots[id];
    if (s.ptr) return RFX_OK;
    const size_t bytes = (size_t)s.rows * s.width * s.texel;
    hipError_t e = hipMalloc(&s.ptr, bytes);
    if (e != hipSuccess) return fail(c, RFX_ENOMEM, "hipMalloc(texture)", e);
    s.owned = true;
    // render targets start zeroed: `discard`ed fragments expose the initial contents (Appendix D-10)
    e = hipMemsetAsync(s.ptr, 0, bytes, c->stream);
    if (e != hipSuccess) return fail(c, RFX_EDEVICE, "hipMemsetAsync", e);
    return RFX_OK;
}

static int band_check(rfx_ctx *c, int id, int row0, int rows) {
    if (id < 0 || id >= RFX_TEX_COUNT) return fail(c, RFX_EINVAL, "bad texture id");
    const Slot &s = c->slots[id];
    if (rows <= 0 || row0 < s.row0 || row0 + rows > s.row0 + s.rows) return fail(c, RFX_EINVAL, "row band outside the rows this context holds");
    return RFX_OK;
}

int rfx_upload(rfx_ctx *c, rfx_tex id, const void *host, int row0, int rows) {
    if (!c || !host) return RFX_EINVAL;
    int rc = band_check(c, id, row0, rows);
    if (rc) return rc;
    if ((rc = ensure(c, id))) return rc;
    RFX_ENTER(c);
    Slot &s = c->slots[id];
    const size_t pitch = (size_t)s.width * s.texel;
    HIPCHK(c, hipMemcpyAsync((char *)s.ptr + (size_t)(row0 - s.row0) * pitch, host, (size_t)rows * pitch, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the caller may free `host` as soon as we return
    s.uploaded = true;
    if (id == RFX_TEX_DEPTH) c->depth_event_set = false;  // complete: nothing for the depth pre-pass to wait for
    return RFX_OK;
}

int rfx_download(rfx_ctx *c, rfx_tex id, void *host, int row0, int rows) {
    if (!c || !host) return RFX_EINVAL;
    int rc = band_check(c, id, row0, rows);
    if (rc) return rc;
    if ((rc = ensure(c, id))) return rc;
    RFX_ENTER(c);
    Slot &s = c->slots[id];
    const size_t pitch = (size_t)s.width * s.texel;
    HIPCHK(c, hipMemcpyAsync(host, (char *)s.ptr + (size_t)(row0 - s.row0) * pitch, (size_t)rows * pitch, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RFX_OK;
}

// ---- streaming dumps: the next frame's planes cross PCIe on their own stream while the current frame is drawn
void *rfx_host_alloc(size_t bytes) {
    void *p = nullptr;
    return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}
void rfx_host_free(void *p) {
    if (p) hipHostFree(p);
}

static bool is_dump_input(int id) { return id == RFX_TEX_DEPTH || id == RFX_TEX_GBUFFER || id == RFX_TEX_VELOCITY || id == RFX_TEX_DIRECT_LIGHT; }

int rfx_stage_upload(rfx_ctx *c, rfx_tex id, const void *host, int row0, int rows) {
    if (!c || !host) return RFX_EINVAL;
    if (!is_dump_input(id)) return fail(c, RFX_EINVAL, "rfx_stage_upload: only the dump's input planes (depth, gbuffer, velocity, direct light) are double-buffered");
    int rc = band_check(c, id, row0, rows);
    if (rc) return rc;
    if ((rc = ensure(c, id))) return rc;
    RFX_ENTER(c);
    Slot &s = c->slots[id];
    if (!s.owned) return fail(c, RFX_ESTATE, "rfx_stage_upload: the slot is bound to an external buffer");
    const size_t pitch = (size_t)s.width * s.texel, bytes = (size_t)s.rows * pitch;
    if (!c->upload_stream) {
        hipError_t e = hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_staged, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_frame_done, hipEventDisableTiming);
        for (hipEvent_t &b : c->ev_batch)
            if (e == hipSuccess) e = hipEventCreateWithFlags(&b, hipEventDisableTiming);
        if (e != hipSuccess) return fail(c, RFX_EDEVICE, "rfx_stage_upload: stream/event creation", e);
        // nothing of an earlier frame can still be reading a back buffer: there is none yet
        HIPCHK(c, hipEventRecord(c->ev_frame_done, c->stream));
    }
    if (!s.back) {
        hipError_t e = hipMalloc(&s.back, bytes);
        if (e != hipSuccess) return fail(c, RFX_ENOMEM, "hipMalloc(back buffer)", e);
    }
    // the buffer being filled was the FRONT buffer until the last flip: the draws that read it were enqueued before that flip
    HIPCHK(c, hipStreamWaitEvent(c->upload_stream, c->ev_frame_done, 0));
    HIPCHK(c, hipMemcpyAsync((char *)s.back + (size_t)(row0 - s.row0) * pitch, host, (size_t)rows * pitch, hipMemcpyHostToDevice, c->upload_stream));
    s.back_filled = true;
    return RFX_OK;
}

int rfx_stage_flip(rfx_ctx *c) {
    if (!c) return RFX_EINVAL;
    if (!c->upload_stream) return fail(c, RFX_ESTATE, "rfx_stage_flip: nothing staged");
    RFX_ENTER(c);
    // draws enqueued from now on wait for the staged copies; copies staged from now on wait for the draws enqueued so far
    HIPCHK(c, hipEventRecord(c->ev_staged, c->upload_stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_staged, 0));
    HIPCHK(c, hipEventRecord(c->ev_frame_done, c->stream));
    // Host-side back pressure.  Neither rfx_stage_upload nor the flip waits for the GPU, and a staged copy executes only once the
    // draws of two frames earlier have finished — a host running ahead (two alternating sets of pinned planes) could refill a set
    // before the copy that reads it has run.  So this flip returns only when the copies published by the PREVIOUS flip have
    // executed: the planes staged before that flip are free to be rewritten, and the host is never more than two frames ahead.
    const unsigned int k = c->flips++;
    HIPCHK(c, hipEventRecord(c->ev_batch[k & 1], c->upload_stream));
    if (k >= 1) HIPCHK(c, hipEventSynchronize(c->ev_batch[(k - 1) & 1]));
    for (int id = 0; id < RFX_TEX_COUNT; id++) {
        Slot &s = c->slots[id];
        if (!s.back_filled) continue;
        if (id == RFX_TEX_DEPTH) {  // the depth pre-pass of the next K1 waits for this copy on its own stream
            HIPCHK(c, hipEventRecord(c->ev_depth, c->upload_stream));
            c->depth_event_set = true;
        }
        void *t = s.ptr; s.ptr = s.back; s.back = t;
        s.back_filled = false;
        s.uploaded = true;
    }
    return RFX_OK;
}

int rfx_clear(rfx_ctx *c, rfx_tex id) {
    if (!c || id < 0 || id >= RFX_TEX_COUNT) return RFX_EINVAL;
    int rc = ensure(c, id);
    if (rc) return rc;
    RFX_ENTER(c);
    Slot &s = c->slots[id];
    HIPCHK(c, hipMemsetAsync(s.ptr, 0, (size_t)s.rows * s.width * s.texel, c->stream));
    if (id == RFX_TEX_DEPTH) {
        HIPCHK(c, hipEventRecord(c->ev_depth, c->stream));
        c->depth_event_set = true;
    }
    return RFX_OK;
}

void *rfx_tex_device_ptr(rfx_ctx *c, rfx_tex id) {
    if (!c || id < 0 || id >= RFX_TEX_COUNT) return nullptr;
    hipSetDevice(c->device);
    if (ensure(c, id)) return nullptr;
    // whoever takes the depth plane's address may write it with work this library cannot see (ordered against the draw stream only, as a
    // bound external buffer is): the pre-pass then stays in the draw stream
    if (id == RFX_TEX_DEPTH) c->depth_external = true;
    return c->slots[id].ptr;
}

int rfx_bind_external(rfx_ctx *c, rfx_tex id, void *device_ptr) {
    if (!c || id < 0 || id >= RFX_TEX_COUNT || !device_ptr) return RFX_EINVAL;
    RFX_ENTER(c);
    Slot &s = c->slots[id];
    if (s.owned && s.ptr) {  // launches that still use the old buffer finish first
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(s.ptr);
    }
    s.ptr = device_ptr;
    s.owned = false;
    s.uploaded = true;
    if (id == RFX_TEX_DEPTH) c->depth_external = true;  // written by whoever owns the buffer, ordered against the draw stream only
    return RFX_OK;
}

static TexView view(rfx_ctx *c, int id) {
    TexView v;
    v.ptr = c->slots[id].ptr; v.row0 = c->slots[id].row0; v.rows = c->slots[id].rows;
    return v;
}
static TexViewW wview(rfx_ctx *c, int id) {
    TexViewW v;
    v.ptr = c->slots[id].ptr; v.row0 = c->slots[id].row0; v.rows = c->slots[id].rows;
    return v;
}
// The plane equations of a w x h render target's vUv (rfx_device.h UvPlanes; this file is compiled with -ffp-contract=off: every product
// below is rounded on its own, as the reference GL's triangle setup rounds them)
static UvPlanes rfx_uv_planes(int model, int w, int h) {
    UvPlanes q;
    q.model = model; q.W = w; q.H = h; q.fW = (float)w; q.fH = (float)h;
    const float ooa = 1.0f / ((float)w * (float)h);
    q.du = (float)h * ooa;
    q.dv = (float)w * ooa;
    const float far_u = q.du * ((float)w - 0.5f), far_v = q.dv * ((float)h - 0.5f);
    q.u0_upper = 0.5f * q.du;
    q.u0_lower = 1.0f - far_u;
    q.v0 = 1.0f - far_v;
    return q;
}
static FrameDims dims(rfx_ctx *c) {
    FrameDims d;
    d.W = c->W; d.H = c->H; d.fW = (float)c->W; d.fH = (float)c->H;
    d.uv = rfx_uv_planes(c->uv_model, c->W, c->H);
    d.halo_violations = c->halo_violations;
    return d;
}
static int need(rfx_ctx *c, const int *ids, int n) {
    for (int i = 0; i < n; i++) {
        int rc = ensure(c, ids[i]);
        if (rc) return rc;
    }
    return RFX_OK;
}

// blue_noise.glsl:9-34: one pcg4d round of the per-draw seed; the toroidal shift is pixel-independent
static void blue_noise_shift(int index, int *sx, int *sy) {
    if (index == 0) { *sx = 0; *sy = 0; return; }  // :38-39 texture-coordinate path == unshifted table
    uint32_t i = (uint32_t)index;
    uint32_t v[4] = {i, i * 15843u, i * 31u + 4566u, i * 2345u + 58585u};
    for (int k = 0; k < 4; k++) v[k] = v[k] * 1664525u + 1013904223u;
    v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
    for (int k = 0; k < 4; k++) v[k] ^= v[k] >> 16;
    v[0] += v[1] * v[3]; v[1] += v[2] * v[0]; v[2] += v[0] * v[1]; v[3] += v[1] * v[2];
    *sx = (int)((v[0] % 0x0fffffffu) % 128u);
    *sy = (int)((v[1] % 0x0fffffffu) % 128u);
}

// Rows a launch produces: the tile, widened by `extra` rows on each side (clipped to what the
// output slot holds).
static bool launch_rows(rfx_ctx *c, int out_id, int extra, int *y0, int *y1) {
    const Slot &s = c->slots[out_id];
    int a = c->tile_y0 - extra, b = c->tile_y0 + c->tile_rows + extra;
    if (a < s.row0) a = s.row0;
    if (b > s.row0 + s.rows) b = s.row0 + s.rows;
    if (a < c->win_y0) a = c->win_y0;  // rfx_set_row_window
    if (b > c->win_y1) b = c->win_y1;
    *y0 = a; *y1 = b;
    return b > a;  // false: nothing to draw
}

// stage `n` host planes (floats per texel in `ch`) of a band on the device, back to back; returns the device base in *stage
static int stage_planes(rfx_ctx *c, const float *const *host, const int *ch, int n, size_t texels, float **stage, const float **dev) {
    size_t total = 0;
    for (int i = 0; i < n; i++) total += host[i] ? texels * ch[i] : 0;
    hipError_t e = hipMalloc((void **)stage, total * sizeof(float));
    if (e != hipSuccess) return fail(c, RFX_ENOMEM, "hipMalloc(AOV staging)", e);
    size_t off = 0;
    for (int i = 0; i < n; i++) {
        dev[i] = nullptr;
        if (!host[i]) continue;
        dev[i] = *stage + off;
        e = hipMemcpyAsync(*stage + off, host[i], texels * ch[i] * sizeof(float), hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { hipFree(*stage); return fail(c, RFX_EDEVICE, "hipMemcpyAsync(AOV plane)", e); }
        off += texels * ch[i];
    }
    return RFX_OK;
}

int rfx_pack_gbuffer(rfx_ctx *c, const rfx_aov_gbuffer *a, int row0, int rows) {
    if (!c || !a || !a->diffuse || !a->normal || !a->roughness || !a->metalness || !a->emissive) return RFX_EINVAL;
    int rc = band_check(c, RFX_TEX_GBUFFER, row0, rows);
    if (rc) return rc;
    if ((rc = ensure(c, RFX_TEX_GBUFFER))) return rc;
    RFX_ENTER(c);
    const float *host[6] = {a->diffuse, a->normal, a->roughness, a->metalness, a->emissive, a->depth}, *dev[6];
    const int ch[6] = {4, 3, 1, 1, 3, 1};
    float *stage = nullptr;
    if ((rc = stage_planes(c, host, ch, 6, (size_t)rows * c->W, &stage, dev))) return rc;
    Slot &s = c->slots[RFX_TEX_GBUFFER];
    hipError_t e = rfx_launch_pack_gbuffer(c->W, rows, dev[0], dev[1], dev[2], dev[3], dev[4], dev[5],
                                           (char *)s.ptr + (size_t)(row0 - s.row0) * s.width * s.texel, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // the caller may free the planes as soon as we return
    hipFree(stage);
    if (e != hipSuccess) return fail(c, RFX_EDEVICE, "rfx_pack_gbuffer", e);
    s.uploaded = true;
    return RFX_OK;
}

int rfx_pack_velocity(rfx_ctx *c, const rfx_aov_velocity *a, int row0, int rows) {
    if (!c || !a || !a->velocity || !a->normal || !a->depth) return RFX_EINVAL;
    int rc = band_check(c, RFX_TEX_VELOCITY, row0, rows);
    if (rc) return rc;
    if ((rc = ensure(c, RFX_TEX_VELOCITY))) return rc;
    RFX_ENTER(c);
    const float *host[3] = {a->velocity, a->normal, a->depth}, *dev[3];
    const int ch[3] = {2, 3, 1};
    float *stage = nullptr;
    if ((rc = stage_planes(c, host, ch, 3, (size_t)rows * c->W, &stage, dev))) return rc;
    Slot &s = c->slots[RFX_TEX_VELOCITY];
    hipError_t e = rfx_launch_pack_velocity(c->W, rows, dev[0], dev[1], dev[2], (char *)s.ptr + (size_t)(row0 - s.row0) * s.width * s.texel, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(stage);
    if (e != hipSuccess) return fail(c, RFX_EDEVICE, "rfx_pack_velocity", e);
    s.uploaded = true;
    return RFX_OK;
}

int rfx_set_environment(rfx_ctx *c, const float *rgba, int width, int height, int halfFloatType, int halfStoreRTZ) {
    if (!c) return RFX_EINVAL;
    RFX_ENTER(c);
    if (!rgba) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->env) hipFree(c->env);
        c->env = nullptr; c->env_w = c->env_h = c->env_levels = 0;
        if (c->env_marginal) hipFree(c->env_marginal);
        if (c->env_conditional) hipFree(c->env_conditional);
        c->env_marginal = c->env_conditional = nullptr;
        return RFX_OK;
    }
    if (width < 1 || height < 1 || width > 16384 || height > 16384 || (width & (width - 1)) || (height & (height - 1)))
        return fail(c, RFX_EINVAL, "rfx_set_environment: width and height must be powers of two <= 16384");
    int levels = 0;
    size_t total = 0;
    unsigned int off[16];
    for (int w = width, h = height;; w = w > 1 ? w >> 1 : 1, h = h > 1 ? h >> 1 : 1) {
        off[levels++] = (unsigned int)total;
        total += (size_t)w * h;
        if (w == 1 && h == 1) break;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->env) hipFree(c->env);
    c->env = nullptr; c->env_w = c->env_h = c->env_levels = 0;
    if (c->env_marginal) hipFree(c->env_marginal);  // tables of the previous map: a new one needs its own
    if (c->env_conditional) hipFree(c->env_conditional);
    c->env_marginal = c->env_conditional = nullptr;
    hipError_t e = hipMalloc((void **)&c->env, total * sizeof(float4));
    if (e != hipSuccess) return fail(c, RFX_ENOMEM, "hipMalloc(environment)", e);
    // staging copy of the base level, then level 0 = the texels in the texture's type, then the chain
    float4 *stage = nullptr;
    e = hipMalloc((void **)&stage, (size_t)width * height * sizeof(float4));
    if (e != hipSuccess) {
        hipFree(c->env);
        c->env = nullptr; c->env_w = c->env_h = c->env_levels = 0;
        return fail(c, RFX_ENOMEM, "hipMalloc(environment staging)", e);
    }
    e = hipMemcpyAsync(stage, rgba, (size_t)width * height * sizeof(float4), hipMemcpyHostToDevice, c->stream);
    // level 0: same size "reduction" = a copy through the type conversion (RNE: the upload of a float image into a half texture)
    if (e == hipSuccess) e = rfx_launch_env_mip(stage, c->env, width, height, width, height, halfFloatType != 0, false, c->stream);
    for (int l = 1, w = width, h = height; l < levels && e == hipSuccess; l++) {
        const int dw = w > 1 ? w >> 1 : 1, dh = h > 1 ? h >> 1 : 1;
        e = rfx_launch_env_mip(c->env + off[l - 1], c->env + off[l], w, h, dw, dh, halfFloatType != 0, halfStoreRTZ != 0, c->stream);
        w = dw; h = dh;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // the caller may free `rgba` as soon as we return
    hipFree(stage);
    if (e != hipSuccess) {
        hipFree(c->env);
        c->env = nullptr; c->env_w = c->env_h = c->env_levels = 0;
        return fail(c, RFX_EDEVICE, "rfx_set_environment: building the mip chain", e);
    }
    c->env_w = width; c->env_h = height; c->env_levels = levels;
    memcpy(c->env_off, off, sizeof off);
    return RFX_OK;
}

int rfx_cube_to_equirect(rfx_ctx *c, const float *faces, int size, int generateMipmaps, float *equirect, int width, int height) {
    if (!c || !faces || !equirect) return RFX_EINVAL;
    if (size < 1 || size > 8192 || width < 1 || height < 1 || width > 16384 || height > 16384)
        return fail(c, RFX_EINVAL, "rfx_cube_to_equirect: face size must be 1..8192, the target 1..16384 in each edge");
    RFX_ENTER(c);
    int levels = 1;
    size_t nchain = (size_t)6 * size * size;
    if (generateMipmaps)
        for (int s = size >> 1; s >= 1; s >>= 1) { nchain += (size_t)6 * s * s; levels++; }
    const size_t nin = (size_t)6 * size * size, nout = (size_t)width * height;
    float4 *din = nullptr, *dout = nullptr;
    hipError_t e = hipMalloc((void **)&din, nchain * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc((void **)&dout, nout * sizeof(float4));
    if (e != hipSuccess) {
        if (din) hipFree(din);
        return fail(c, RFX_ENOMEM, "hipMalloc(cube chain / equirect target)", e);
    }
    e = hipMemcpyAsync(din, faces, nin * sizeof(float4), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = rfx_launch_cube_to_equirect(din, size, levels, dout, width, height, rfx_uv_planes(c->uv_model, width, height), c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(equirect, dout, nout * sizeof(float4), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(din);
    hipFree(dout);
    if (e != hipSuccess) return fail(c, RFX_EDEVICE, "rfx_cube_to_equirect", e);
    return RFX_OK;
}

int rfx_set_environment_importance(rfx_ctx *c, const float *marginal, size_t marginalCount, const float *conditional, size_t conditionalCount,
                                   float totalSumWhole, float totalSumDecimal) {
    if (!c || !marginal || !conditional) return RFX_EINVAL;
    if (!c->env) return fail(c, RFX_ESTATE, "rfx_set_environment_importance: no environment set");
    if (marginalCount != (size_t)c->env_h || conditionalCount != (size_t)c->env_w * c->env_h)
        return fail(c, RFX_EINVAL, "rfx_set_environment_importance: marginalWeights must hold height floats and conditionalWeights width*height");
    RFX_ENTER(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->env_marginal) hipFree(c->env_marginal);
    if (c->env_conditional) hipFree(c->env_conditional);
    c->env_marginal = c->env_conditional = nullptr;
    hipError_t e = hipMalloc((void **)&c->env_marginal, (size_t)c->env_h * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&c->env_conditional, (size_t)c->env_w * c->env_h * sizeof(float));
    if (e != hipSuccess) {
        if (c->env_marginal) hipFree(c->env_marginal);
        c->env_marginal = c->env_conditional = nullptr;
        return fail(c, RFX_ENOMEM, "hipMalloc(environment importance tables)", e);
    }
    e = hipMemcpyAsync(c->env_marginal, marginal, (size_t)c->env_h * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->env_conditional, conditional, (size_t)c->env_w * c->env_h * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {  // tables with undefined contents must not pass the importanceSampling validation
        hipFree(c->env_marginal); hipFree(c->env_conditional);
        c->env_marginal = c->env_conditional = nullptr;
        return fail(c, RFX_EDEVICE, "rfx_set_environment_importance: copying the tables", e);
    }
    c->env_sum_whole = totalSumWhole; c->env_sum_decimal = totalSumDecimal;
    return RFX_OK;
}

int rfx_download_environment(rfx_ctx *c, int level, float *rgba, int *levels) {
    if (!c) return RFX_EINVAL;
    if (levels) *levels = c->env_levels;
    if (!rgba) return RFX_OK;
    if (!c->env || level < 0 || level >= c->env_levels) return fail(c, RFX_EINVAL, "rfx_download_environment: no such level");
    const int w = (c->env_w >> level) > 0 ? c->env_w >> level : 1, h = (c->env_h >> level) > 0 ? c->env_h >> level : 1;
    RFX_ENTER(c);
    HIPCHK(c, hipMemcpyAsync(rgba, c->env + c->env_off[level], (size_t)w * h * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RFX_OK;
}

// stage 0: rfx_ssgi_march (one launch); 1: rfx_ssgi_trace; 2: rfx_ssgi_shade
static int ssgi_draw(rfx_ctx *c, const rfx_ssgi_params *p, int stage) {
    if (!c || !p) return RFX_EINVAL;
    if (p->mode != 0 && p->mode != 1) return fail(c, RFX_EINVAL, "rfx_ssgi_march/trace/shade: mode must be 0 (MODE_SSGI) or 1 (MODE_SSR)");
    if (p->importanceSampling && (!p->useEnvMap || !c->env_marginal))
        return fail(c, RFX_ESTATE, "rfx_ssgi_march/trace/shade: importanceSampling needs useEnvMap and rfx_set_environment_importance");
    if (p->useEnvMap && !c->env) return fail(c, RFX_ESTATE, "rfx_ssgi_march/trace/shade: useEnvMap without rfx_set_environment");
    if (p->steps < 1 || p->refineSteps < 0) return fail(c, RFX_EINVAL, "rfx_ssgi_march/trace/shade: steps/refineSteps");
    RFX_ENTER(c);
    if (p->historySource < 0 || p->historySource > 3) return fail(c, RFX_EINVAL, "rfx_ssgi_march/trace/shade: historySource");
    if (p->historySource == 1 && (c->tile_y0 != 0 || c->tile_rows != c->H))
        return fail(c, RFX_EUNSUPPORTED, "rfx_ssgi_march/trace/shade: historySource TEMPORAL0 (denoiseMode \"temporal\") needs a whole-frame context: K1 gathers it anywhere on screen");
    const int hist = p->historySource == 1 ? RFX_TEX_TEMPORAL0 : (p->historySource == 3 ? RFX_TEX_COMPOSE_RGB : RFX_TEX_COMPOSE);
    const int ids[] = {RFX_TEX_DEPTH, RFX_TEX_GBUFFER, RFX_TEX_DIRECT_LIGHT, hist, RFX_TEX_BLUE_NOISE, RFX_TEX_SSGI};
    int rc = need(c, ids, 6);
    if (rc) return rc;
    if (!c->slots[RFX_TEX_DEPTH].uploaded || !c->slots[RFX_TEX_GBUFFER].uploaded || !c->slots[RFX_TEX_BLUE_NOISE].uploaded)
        return fail(c, RFX_ESTATE, "rfx_ssgi_march/trace/shade: depth / gbuffer / blue-noise not uploaded");
    K1Args A;
    A.dims = dims(c);
    // K2's neighbourhood clamp reads +-2 rows of K1's output: produce them redundantly in the halo
    bool any = launch_rows(c, RFX_TEX_SSGI, c->halo < 2 ? c->halo : 2, &A.y0, &A.y1);
    A.out_w = c->W; A.out_h = c->H;
    const float rs = p->resolutionScale == 0.0f ? 1.0f : p->resolutionScale;
    if (rs != 1.0f) {  // SSGIPass.setSize :52-57
        const float fw = (float)c->W * rs, fh = (float)c->H * rs;
        if (!(rs > 0.0f && rs <= 1.0f) || fw != floorf(fw) || fh != floorf(fh) || fw < 1.0f || fh < 1.0f)
            return fail(c, RFX_EINVAL, "rfx_ssgi_march/trace/shade: resolutionScale must be in (0, 1] with whole W*s and H*s");
        if (c->tile_y0 != 0 || c->tile_rows != c->H) return fail(c, RFX_EUNSUPPORTED, "rfx_ssgi_march/trace/shade: resolutionScale != 1 needs a whole-frame context");
        A.out_w = (int)fw; A.out_h = (int)fh;
        A.y0 = 0; A.y1 = A.out_h;
        any = true;  // whole-frame contexts only: the row window does not apply to the scaled target
    }
    A.out_uv = rfx_uv_planes(c->uv_model, A.out_w, A.out_h);
    A.depth = view(c, RFX_TEX_DEPTH); A.gbuffer = view(c, RFX_TEX_GBUFFER); A.direct = view(c, RFX_TEX_DIRECT_LIGHT);
    A.history = view(c, hist);
    A.blue = c->slots[RFX_TEX_BLUE_NOISE].ptr;
    blue_noise_shift(p->blueNoiseIndex, &A.shift_x, &A.shift_y);
    A.out = wview(c, RFX_TEX_SSGI);
    A.p = *p;
    // SSGIPass.js:84-87: computed in JS doubles, then rounded to float uniforms
    A.nearMulFar = (float)((double)p->camera.near_ * (double)p->camera.far_);
    A.farMinusNear = (float)((double)p->camera.far_ - (double)p->camera.near_);
    A.nearMinusFar = (float)((double)p->camera.near_ - (double)p->camera.far_);
    const int base = rfx_k1_base_cell();
    A.coarse_w = (c->W + base - 1) / base;
    A.coarse_h = (c->H + base - 1) / base;
    // the march's table lives in every workgroup's LDS (four workgroups per CU): the cell edge is doubled until the table fits.  Two layouts
    // (k1_tap_at): rows padded to a power of two, at least 2^cell_shift cells — a tap's LDS address is then two shifts and one v_bitop3_b32 — when
    // that fits the 36 KiB at the SAME cell size as plain rows of cells_w cells would (4K: 32-texel cells, 128 x 68 cells = 34 KiB; every 16:9
    // frame); plain rows otherwise (an ultrawide frame, a frame taller than 9216 rows: the padded table of such a frame holds at least H cells)
    const int table_budget = 36864;
    const auto k1_table = [&](int shift, bool pow2, int &cw, int &ch, int &pitch, int &pitch_log2) {
        cw = (c->W + (1 << shift) - 1) >> shift;
        ch = (c->H + (1 << shift) - 1) >> shift;
        pitch = cw;
        pitch_log2 = 0;
        if (pow2) {
            pitch_log2 = shift;  // (at least 2^cell_shift cells per row: the row term of k1_tap_at is then a LEFT shift by >= 2)
            while ((1 << pitch_log2) < cw) pitch_log2++;
            pitch = 1 << pitch_log2;
        }
        return (size_t)((pitch * ch + 3) / 4) * 16;  // bytes, whole uint4s
    };
    A.cells_pow2 = 0;
    for (A.cell_shift = 4;; A.cell_shift++) {
        if (k1_table(A.cell_shift, false, A.cells_w, A.cells_h, A.cells_pitch, A.cells_pitch_log2) <= (size_t)table_budget || A.cell_shift >= 12) break;
    }
    if (RFX_K1_POW2) {
        int cw, ch, pitch, pl2;
        if (k1_table(A.cell_shift, true, cw, ch, pitch, pl2) <= (size_t)table_budget) {
            A.cells_pow2 = 1;
            A.cells_pitch = pitch;
            A.cells_pitch_log2 = pl2;
        }
    }
    A.cells_vec4 = (A.cells_pitch * A.cells_h + 3) / 4;
    if (!c->viewz) {
        hipError_t e = hipMalloc((void **)&c->viewz, (size_t)c->W * c->H * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void **)&c->coarse, (size_t)A.coarse_w * A.coarse_h * sizeof(float2));
        if (e == hipSuccess) e = hipMalloc((void **)&c->cells, (size_t)A.cells_vec4 * 16);
        if (e == hipSuccess) e = hipMalloc((void **)&c->k1_tiles, 64 * 128);
        if (e != hipSuccess) {  // all four or none: a later draw must not find viewz set and the tables missing
            if (c->viewz) hipFree(c->viewz);
            if (c->coarse) hipFree(c->coarse);
            if (c->cells) hipFree(c->cells);
            if (c->k1_tiles) hipFree(c->k1_tiles);
            c->viewz = nullptr; c->coarse = nullptr; c->cells = nullptr; c->k1_tiles = nullptr;
            return fail(c, RFX_ENOMEM, "hipMalloc(K1 scratch)", e);
        }
    }
    A.viewz = c->viewz;
    A.coarse = c->coarse;
    A.cells = c->cells;
    A.tile_counter = c->k1_tiles;
    A.n_cu = c->n_cu;
    A.env = c->env;
    A.env_w = c->env_w; A.env_h = c->env_h; A.env_levels = c->env_levels;
    A.env_marginal = c->env_marginal; A.env_conditional = c->env_conditional;
    A.totalSumWhole = c->env_sum_whole; A.totalSumDecimal = c->env_sum_decimal;
    memcpy(A.env_off, c->env_off, sizeof A.env_off);
    {   // getMaxMipLevel (src/ssgi/utils/Utils.js:30-34): floor(log2(max(w, h))) + 1
        int m = c->env_w > c->env_h ? c->env_w : c->env_h, lg = 0;
        while ((m >> (lg + 1)) > 0) lg++;
        A.maxEnvMapMipLevel = c->env ? (float)(lg + 1) : 0.0f;
    }
    A.hits = nullptr;
    if (stage != 0) {
        // hand-over plane, indexed like the output texture (resolutionScale needs a whole-frame context, so W * held rows covers it)
        const size_t n = (size_t)c->W * c->slots[RFX_TEX_SSGI].rows * 2;
        if (stage == 2 && (!c->hits || !c->hits_traced))
            return fail(c, RFX_ESTATE, "rfx_ssgi_shade: no rfx_ssgi_trace of this frame to finish");
        if (!c->hits) {
            hipError_t e = hipMalloc((void **)&c->hits, n * sizeof(float4));
            if (e != hipSuccess) return fail(c, RFX_ENOMEM, "hipMalloc(K1 trace hand-over)", e);
        }
        A.hits = c->hits;
    }
    // the pre-pass runs on EVERY draw: the depth plane is an input that changes every frame (the shade stage reuses the trace's).
    // On its own stream (rfx_ctx.h prep_stream) unless the depth plane lives in a caller's buffer: after the depth plane's last writer and
    // after the previous K1 launch (which read the scratch planes), NOT after the draws queued since — it overlaps them.
    if (stage != 2) {
#ifndef RFX_K1_PREP_STREAM
#define RFX_K1_PREP_STREAM 1  // build knob: 0 = the pre-pass in the draw stream (A/B measurements)
#endif
        if (c->depth_external || !RFX_K1_PREP_STREAM) {
            ProfScope prof(c, RFX_PROF_K1_PREPASS, c->stream);
            HIPCHK(c, rfx_launch_k1_prepare(A, c->stream));
        } else {
            if (c->depth_event_set) HIPCHK(c, hipStreamWaitEvent(c->prep_stream, c->ev_depth, 0));
            if (c->k1_event_set) HIPCHK(c, hipStreamWaitEvent(c->prep_stream, c->ev_k1_done, 0));
            {
                ProfScope prof(c, RFX_PROF_K1_PREPASS, c->prep_stream);
                HIPCHK(c, rfx_launch_k1_prepare(A, c->prep_stream));
            }
            HIPCHK(c, hipEventRecord(c->ev_prep_done, c->prep_stream));
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_prep_done, 0));
        }
    }
    // the march kernel hands its tiles out from a counter: the pre-pass zeroes it; the shade stage has no pre-pass of its own
    if (any && stage == 2) HIPCHK(c, hipMemsetAsync(c->k1_tiles, 0, 64 * 128, c->stream));
    if (any) {
        ProfScope prof(c, RFX_PROF_K1_MARCH, c->stream);
        HIPCHK(c, rfx_launch_k1(A, stage, c->stream));
    }
    HIPCHK(c, hipEventRecord(c->ev_k1_done, c->stream));  // the next pre-pass overwrites what this launch reads
    c->k1_event_set = true;
    c->hits_traced = stage == 1;
    if (stage == 1) { c->trace_y0 = A.y0; c->trace_y1 = any ? A.y1 : A.y0; c->trace_missed = p->missedRays; c->trace_scaled = rs != 1.0f; }
    return RFX_OK;
}

// between rfx_ssgi_trace and rfx_ssgi_shade (rfx_gather_history_rows): which rows of last frame's composed GI will the shade read?
int rfx_internal_hit_rows_enqueue(rfx_ctx *c, int *rows_dev) {
    if (!c->hits || !c->hits_traced) return fail(c, RFX_ESTATE, "rfx_gather_history_rows: no rfx_ssgi_trace of this frame is waiting for its shade");
    // the hand-over plane of a resolutionScale != 1 trace is indexed by the SMALLER target (and such a trace needs a whole-frame context, which
    // has no history to gather): the row reduction below reads it with the frame's pitch
    if (c->trace_scaled) return fail(c, RFX_EUNSUPPORTED, "rfx_ssgi_hit_rows / rfx_gather_history_rows: the last rfx_ssgi_trace ran with resolutionScale != 1");
    RFX_ENTER(c);
    static const int preset[2] = {0x7fffffff, -1};
    HIPCHK(c, hipMemcpyAsync(rows_dev, preset, sizeof preset, hipMemcpyHostToDevice, c->stream));
    if (c->trace_y1 > c->trace_y0)
        HIPCHK(c, rfx_launch_k1_hit_rows(dims(c), c->trace_y0, c->trace_y1, view(c, RFX_TEX_DEPTH), wview(c, RFX_TEX_SSGI), c->hits, c->trace_missed != 0, rows_dev, c->stream));
    return RFX_OK;
}

int rfx_internal_hit_mask_enqueue(rfx_ctx *c, int ranks) {
    if (!c->hits || !c->hits_traced) return fail(c, RFX_ESTATE, "rfx_gather_history_rows / rfx_ssgi_hit_mask: no rfx_ssgi_trace of this frame is waiting for its shade");
    if (c->trace_scaled) return fail(c, RFX_EUNSUPPORTED, "rfx_ssgi_hit_mask / rfx_gather_history_rows: the last rfx_ssgi_trace ran with resolutionScale != 1");
    RFX_ENTER(c);
    if (ranks < 1) ranks = 1;
    if (!c->hit_mask_dev || c->hit_mask_ranks < ranks) {
        if (c->hit_mask_dev) {
            // the packing kernels and the offset copy of an earlier rfx_gather_history_rows run on comm_stream and read these buffers (the row
            // offsets live in the same allocation): both streams drain before they go
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (c->comm_stream) HIPCHK(c, hipStreamSynchronize(c->comm_stream));
            hipFree(c->hit_mask_dev); hipHostFree(c->hit_mask_host); c->hit_mask_dev = nullptr; c->hit_mask_host = nullptr;
        }
        const size_t words = (size_t)(2 * ranks + 2) * c->H;  // [0, H) this tile's mask, [H, (n + 1) H) every rank's, then (n + 1) H row offsets
        hipError_t e = hipMalloc((void **)&c->hit_mask_dev, words * sizeof(unsigned int));
        if (e == hipSuccess) e = hipHostMalloc((void **)&c->hit_mask_host, words * sizeof(unsigned int), hipHostMallocDefault);
        if (e != hipSuccess) return fail(c, RFX_ENOMEM, "rfx_ssgi_hit_mask: scratch", e);
        c->hit_mask_ranks = ranks;
    }
    HIPCHK(c, hipMemsetAsync(c->hit_mask_dev, 0, (size_t)c->H * sizeof(unsigned int), c->stream));
    if (c->trace_y1 > c->trace_y0)
        HIPCHK(c, rfx_launch_k1_hit_mask(dims(c), c->trace_y0, c->trace_y1, view(c, RFX_TEX_DEPTH), wview(c, RFX_TEX_SSGI), c->hits, c->trace_missed != 0, c->hit_mask_dev, c->stream));
    return RFX_OK;
}

int rfx_ssgi_hit_mask(rfx_ctx *c, unsigned int *row_mask, int rows) {
    if (!c || !row_mask) return RFX_EINVAL;
    if (rows != c->H) return fail(c, RFX_EINVAL, "rfx_ssgi_hit_mask: one word per frame row (rows == height)");
    int rc = rfx_internal_hit_mask_enqueue(c, 1);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->hit_mask_host, c->hit_mask_dev, (size_t)c->H * sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(row_mask, c->hit_mask_host, (size_t)c->H * sizeof(unsigned int));
    return RFX_OK;
}

int rfx_ssgi_hit_rows(rfx_ctx *c, int *row_lo, int *row_hi) {
    if (!c || !row_lo || !row_hi) return RFX_EINVAL;
    RFX_ENTER(c);
    if (!c->hit_rows_dev) {  // sized for any communicator this context may get later: 2 + 2 * 64 ranks
        hipError_t e = hipMalloc((void **)&c->hit_rows_dev, sizeof(int) * 130);
        if (e == hipSuccess) e = hipHostMalloc((void **)&c->hit_rows_host, sizeof(int) * 128, hipHostMallocDefault);
        if (e != hipSuccess) return fail(c, RFX_ENOMEM, "rfx_ssgi_hit_rows: scratch", e);
    }
    int rc = rfx_internal_hit_rows_enqueue(c, c->hit_rows_dev);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->hit_rows_host, c->hit_rows_dev, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *row_lo = c->hit_rows_host[0];
    *row_hi = c->hit_rows_host[1];
    return RFX_OK;
}

int rfx_ssgi_march(rfx_ctx *c, const rfx_ssgi_params *p) { return ssgi_draw(c, p, 0); }
int rfx_ssgi_trace(rfx_ctx *c, const rfx_ssgi_params *p) { return ssgi_draw(c, p, 1); }
int rfx_ssgi_shade(rfx_ctx *c, const rfx_ssgi_params *p) { return ssgi_draw(c, p, 2); }


int rfx_temporal_reproject(rfx_ctx *c, const rfx_temporal_params *p) {
    if (!c || !p) return RFX_EINVAL;
    if (!((p->inputType == 0 && p->textureCount == 2) || ((p->inputType == 1 || p->inputType == 2) && p->textureCount == 1)))
        return fail(c, RFX_EINVAL, "rfx_temporal_reproject: inputType/textureCount combination");
    if (p->historySource < 0 || p->historySource > 2) return fail(c, RFX_EINVAL, "rfx_temporal_reproject: historySource");
    RFX_ENTER(c);
    const int h0 = p->historySource == 0 ? RFX_TEX_DENOISE_B0 : (p->historySource == 1 ? RFX_TEX_FBCOPY_F16 : RFX_TEX_FBCOPY_F32);
    // with one texture the reference binds the same history to every index (TemporalReprojectPass.js:148-151)
    const int h1 = (p->historySource == 0 && p->textureCount == 2) ? RFX_TEX_DENOISE_B1 : h0;
    const int o1 = p->textureCount == 2 ? RFX_TEX_TEMPORAL1 : RFX_TEX_TEMPORAL0;
    const int ids[] = {RFX_TEX_SSGI, RFX_TEX_VELOCITY, h0, h1, RFX_TEX_TEMPORAL0, o1};
    int rc = need(c, ids, 6);
    if (rc) return rc;
    if (!c->slots[RFX_TEX_VELOCITY].uploaded) return fail(c, RFX_ESTATE, "rfx_temporal_reproject: velocity not uploaded");
    K2Args A;
    A.dims = dims(c);
    if (!launch_rows(c, RFX_TEX_TEMPORAL0, 0, &A.y0, &A.y1)) return RFX_OK;
    A.ssgi = view(c, RFX_TEX_SSGI); A.velocity = view(c, RFX_TEX_VELOCITY);
    A.hist0 = view(c, h0);
    A.hist1 = view(c, h1);
    A.hist_f32 = p->historySource == 2;
    A.in_w = p->inputWidth > 0 ? p->inputWidth : c->W;
    A.in_h = p->inputHeight > 0 ? p->inputHeight : c->H;
    if (A.in_w > c->W || A.in_h > c->H) return fail(c, RFX_EINVAL, "rfx_temporal_reproject: inputWidth/inputHeight larger than the frame");
    if ((A.in_w != c->W || A.in_h != c->H) && (c->tile_y0 != 0 || c->tile_rows != c->H))
        return fail(c, RFX_EUNSUPPORTED, "rfx_temporal_reproject: a smaller input texture (resolutionScale != 1) needs a whole-frame context");
    A.out0 = wview(c, RFX_TEX_TEMPORAL0); A.out1 = wview(c, o1);
    A.p = *p;
    // TemporalReprojectPass.js:135: invTexSize.set(1 / width, 1 / height) in doubles
    A.invW = (float)(1.0 / (double)c->W); A.invH = (float)(1.0 / (double)c->H);
    {   // IEEE fp32 reciprocals of those two uniforms (volatile: no folding into double arithmetic)
        volatile float iw = A.invW, ih = A.invH;
        A.rcpInvW = 1.0f / iw; A.rcpInvH = 1.0f / ih;
    }
    // prevProjectionMatrix * prevViewMatrix (reproject.frag:183), fp32, column by column like GLSL
    const float *Pm = p->prevCamera.projectionMatrix, *Vm = p->prevCamera.matrixWorldInverse;
    for (int col = 0; col < 4; col++)
        for (int row = 0; row < 4; row++) {
            volatile float acc = Pm[0 * 4 + row] * Vm[col * 4 + 0];
            volatile float t1 = Pm[1 * 4 + row] * Vm[col * 4 + 1]; acc = acc + t1;
            volatile float t2 = Pm[2 * 4 + row] * Vm[col * 4 + 2]; acc = acc + t2;
            volatile float t3 = Pm[3 * 4 + row] * Vm[col * 4 + 3]; acc = acc + t3;
            A.prevPV[col * 4 + row] = acc;
        }
    ProfScope prof(c, RFX_PROF_K2, c->stream);
    HIPCHK(c, rfx_launch_k2(A, c->stream));
    return RFX_OK;
}

int rfx_copy_framebuffer(rfx_ctx *c, rfx_tex dst) {
    if (!c) return RFX_EINVAL;
    if (dst != RFX_TEX_FBCOPY_F16 && dst != RFX_TEX_FBCOPY_F32) return fail(c, RFX_EINVAL, "rfx_copy_framebuffer: dst must be RFX_TEX_FBCOPY_F16 or _F32");
    RFX_ENTER(c);
    const int ids[] = {RFX_TEX_TEMPORAL0, (int)dst};
    int rc = need(c, ids, 2);
    if (rc) return rc;
    int y0, y1;
    if (!launch_rows(c, dst, 0, &y0, &y1)) return RFX_OK;
    HIPCHK(c, rfx_launch_copy_fb(dims(c), y0, y1, view(c, RFX_TEX_TEMPORAL0), wview(c, dst), dst == RFX_TEX_FBCOPY_F16, c->stream));
    return RFX_OK;
}

int rfx_poisson_denoise(rfx_ctx *c, const rfx_denoise_params *p) {
    if (!c || !p) return RFX_EINVAL;
    if (p->textureCount != 1 && p->textureCount != 2) return fail(c, RFX_EINVAL, "rfx_poisson_denoise: textureCount");
    RFX_ENTER(c);
    const int in0 = p->inputIsTemporal ? RFX_TEX_TEMPORAL0 : (p->writeToB ? RFX_TEX_DENOISE_A0 : RFX_TEX_DENOISE_B0);
    const int in1 = p->inputIsTemporal ? RFX_TEX_TEMPORAL1 : (p->writeToB ? RFX_TEX_DENOISE_A1 : RFX_TEX_DENOISE_B1);
    const int out0 = p->writeToB ? RFX_TEX_DENOISE_B0 : RFX_TEX_DENOISE_A0;
    const int out1 = p->writeToB ? RFX_TEX_DENOISE_B1 : RFX_TEX_DENOISE_A1;
    const int ids[] = {RFX_TEX_DEPTH, RFX_TEX_GBUFFER, RFX_TEX_BLUE_NOISE, in0, in1, out0, out1};
    int rc = need(c, ids, 7);
    if (rc) return rc;
    if (!c->slots[RFX_TEX_DEPTH].uploaded || !c->slots[RFX_TEX_GBUFFER].uploaded || !c->slots[RFX_TEX_BLUE_NOISE].uploaded)
        return fail(c, RFX_ESTATE, "rfx_poisson_denoise: depth / gbuffer / blue-noise not uploaded");
    K3Args A;
    A.dims = dims(c);
    if (!launch_rows(c, out0, 0, &A.y0, &A.y1)) return RFX_OK;
    A.depth = view(c, RFX_TEX_DEPTH); A.gbuffer = view(c, RFX_TEX_GBUFFER);
    A.in0 = view(c, in0);
    A.in1 = view(c, p->textureCount == 2 ? in1 : in0);  // `#define inputTexture2 inputTexture` (poisson_denoise.frag:30-32)
    A.blue = c->slots[RFX_TEX_BLUE_NOISE].ptr;
    blue_noise_shift(p->blueNoiseIndex, &A.shift_x, &A.shift_y);
    A.out0 = wview(c, out0); A.out1 = wview(c, out1);
    A.p = *p;
    ProfScope prof(c, p->inputIsTemporal ? RFX_PROF_K3_PASS0 : RFX_PROF_K3_PASSN, c->stream);
    HIPCHK(c, rfx_launch_k3(A, c->stream));
    return RFX_OK;
}

int rfx_compose(rfx_ctx *c, const rfx_compose_params *p) {
    if (!c || !p) return RFX_EINVAL;
    if (p->inputType != 0 && p->inputType != 2)
        return fail(c, RFX_EUNSUPPORTED, "rfx_compose: inputType diffuseSpecular (0) and specular (2) are built");
    RFX_ENTER(c);
    if (p->giSource != 0 && p->giSource != 1) return fail(c, RFX_EINVAL, "rfx_compose: giSource");
    const int g0 = p->giSource ? RFX_TEX_TEMPORAL0 : RFX_TEX_DENOISE_B0, g1 = p->giSource ? RFX_TEX_TEMPORAL1 : RFX_TEX_DENOISE_B1;
    const int ids[] = {RFX_TEX_DEPTH, RFX_TEX_GBUFFER, g0, g1, RFX_TEX_COMPOSE, RFX_TEX_DIRECT_LIGHT};
    int rc = need(c, ids, 6);
    if (rc) return rc;
    K4Args A;
    A.dims = dims(c);
    launch_rows(c, RFX_TEX_COMPOSE, 0, &A.y0, &A.y1);
    if (A.y0 < c->tile_y0) A.y0 = c->tile_y0;  // COMPOSE is held whole: write only the tile
    if (A.y1 > c->tile_y0 + c->tile_rows) A.y1 = c->tile_y0 + c->tile_rows;
    const bool any = A.y1 > A.y0;
    A.depth = view(c, RFX_TEX_DEPTH); A.gbuffer = view(c, RFX_TEX_GBUFFER);
    A.gi0 = view(c, g0); A.gi1 = view(c, g1);
    A.scene = view(c, RFX_TEX_DIRECT_LIGHT);  // Denoiser.js:101-103: sceneTexture = the composer's input buffer
    A.out = wview(c, RFX_TEX_COMPOSE);
    A.rgb_out = nullptr;
    if (p->writeHistoryRGB) {
        const int rgb[] = {RFX_TEX_COMPOSE_RGB};
        if ((rc = need(c, rgb, 1))) return rc;
        A.rgb_out = (float *)c->slots[RFX_TEX_COMPOSE_RGB].ptr;  // held whole, like COMPOSE: frame row y at y * W
    }
    A.p = *p;
    if (c->peer_release_pending) {  // rfx_peer_gather_history: this draw overwrites rows a peer's kernel may still be pulling
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_peer_release, 0));
        c->peer_release_pending = false;
    }
    if (any) {
        ProfScope prof(c, RFX_PROF_K4, c->stream);
        HIPCHK(c, rfx_launch_k4(A, c->stream));
    }
    return RFX_OK;
}

int rfx_final_compose(rfx_ctx *c, const rfx_final_params *p) {
    if (!c || !p) return RFX_EINVAL;
    if (p->fogMode < 0 || p->fogMode > 2) return fail(c, RFX_EINVAL, "rfx_final_compose: fogMode");
    RFX_ENTER(c);
    if (p->inputSource < 0 || p->inputSource > 2) return fail(c, RFX_EINVAL, "rfx_final_compose: inputSource");
    const int src = p->inputSource == 0 ? RFX_TEX_COMPOSE : (p->inputSource == 1 ? RFX_TEX_TEMPORAL0 : RFX_TEX_DENOISE_B0);
    const int ids[] = {RFX_TEX_DEPTH, src, RFX_TEX_DIRECT_LIGHT, RFX_TEX_FINAL};
    int rc = need(c, ids, 4);
    if (rc) return rc;
    K5Args A;
    A.dims = dims(c);
    if (!launch_rows(c, RFX_TEX_FINAL, 0, &A.y0, &A.y1)) return RFX_OK;
    A.depth = view(c, RFX_TEX_DEPTH); A.gi = view(c, src); A.scene = view(c, RFX_TEX_DIRECT_LIGHT);
    A.out = wview(c, RFX_TEX_FINAL);
    A.p = *p;
    ProfScope prof(c, RFX_PROF_K5, c->stream);
    HIPCHK(c, rfx_launch_k5(A, c->stream));
    return RFX_OK;
}

int rfx_sync(rfx_ctx *c) {
    if (!c) return RFX_EINVAL;
    RFX_ENTER(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return RFX_OK;
}

int rfx_time_begin(rfx_ctx *c) {
    if (!c) return RFX_EINVAL;
    RFX_ENTER(c);
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    return RFX_OK;
}
int rfx_time_end(rfx_ctx *c, float *elapsed_ms) {
    if (!c || !elapsed_ms) return RFX_EINVAL;
    RFX_ENTER(c);
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    HIPCHK(c, hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1));
    return RFX_OK;
}

int rfx_profile(rfx_ctx *c, int enable) {
    if (!c) return RFX_EINVAL;
    RFX_ENTER(c);
    if (enable) {  // events of an earlier run may still be pending: let them execute before they are recorded again
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->prep_stream) HIPCHK(c, hipStreamSynchronize(c->prep_stream));
        prof_recycle(c);
    }
    c->profiling = enable != 0;
    return RFX_OK;
}
int rfx_profile_read(rfx_ctx *c, float *ms_sum, int *launches) {
    if (!c) return RFX_EINVAL;
    RFX_ENTER(c);
    // every pair is waited for on its own: the draws may have been enqueued on a stream that is no longer the current one (rfx_set_stream between
    // the draws and this call), which a synchronisation of today's streams would not cover; a pair that cannot be read is skipped, not fatal
    float ms[RFX_PROF_COUNT] = {0};
    int n[RFX_PROF_COUNT] = {0};
    for (const rfx_ctx::ProfRec &r : c->prof_recs) {
        float t = 0.0f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) {
            (void)hipGetLastError();
            continue;
        }
        ms[r.kind] += t;
        n[r.kind]++;
    }
    for (int i = 0; i < RFX_PROF_COUNT; i++) {
        if (ms_sum) ms_sum[i] = ms[i];
        if (launches) launches[i] = n[i];
    }
    return RFX_OK;
}

unsigned int rfx_halo_violations(rfx_ctx *c) {
    unsigned int v = 0;
    if (!c) return 0;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    hipMemcpy(&v, c->halo_violations, sizeof v, hipMemcpyDeviceToHost);
    return v;
}

}  // extern "C"
