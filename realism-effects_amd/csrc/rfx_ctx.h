// rfx_ctx.h — the context object behind the opaque rfx_ctx handle, shared by rfx_api.hip (slots, draws) and rfx_comm.hip
// (RCCL exchanges of a row-tiled run).  Private to librfx_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string>
#include <vector>
#include "../../include/rfx.h"

struct Slot {
    void *ptr = nullptr;
    bool owned = false;
    int row0 = 0, rows = 0;  // held band (frame rows)
    size_t texel = 0;
    int width = 0;
    bool uploaded = false;
    // streaming dumps (rfx_stage_upload / rfx_stage_flip): the BACK buffer the next frame's plane is copied into while the draws read `ptr`
    void *back = nullptr;
    bool back_filled = false;
};

struct rfx_ctx {
    int device = 0;
    int W = 0, H = 0, tile_y0 = 0, tile_rows = 0, halo = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    unsigned int *halo_violations = nullptr;
    float *viewz = nullptr;    // K1 scratch: view-space Z plane (full frame)
    float4 *hits = nullptr;    // K1 trace -> shade hand-over (rfx_ssgi_trace), 2 texels per SSGI texel
    bool hits_traced = false;  // a trace is waiting for its shade
    int trace_y0 = 0, trace_y1 = 0, trace_missed = 0;  // the rows and the missedRays option of that trace (rfx_gather_history_rows)
    bool trace_scaled = false;                         // ... and whether it drew a smaller target (resolutionScale != 1)
    int *hit_rows_dev = nullptr;   // device: [0..1] this tile's (min, max) needed history row, [2..2n+1] every rank's
    int *hit_rows_host = nullptr;  // pinned mirror of the gathered part
    // the bounded gather's row masks (rfx_gather_history_rows, rfx_ssgi_hit_mask): one word per frame row; device: [0, H) this tile's, [H, (n+1) H) every rank's
    unsigned int *hit_mask_dev = nullptr, *hit_mask_host = nullptr;
    int hit_mask_ranks = 0;  // ranks the two buffers are sized for (each holds (2 n + 2) H words: the masks, then the packed transfer's row offsets)
    void *hist_staging = nullptr;      // the bounded gather's packed messages: what this rank sends, then what it receives
    size_t hist_staging_bytes = 0;
    // K1's depth pre-pass (view-Z plane + (min, max) cells) depends on the frame's depth plane only: it runs on its own stream, after
    // the PREVIOUS frame's K1 (the last reader of the scratch it overwrites) and under that frame's K2 / K3 / K4, which are still queued
    // or executing when the host issues the next frame.  ev_depth: the depth slot's last asynchronous writer (rfx_stage_flip / rfx_clear).
    hipStream_t prep_stream = nullptr;
    hipEvent_t ev_depth = nullptr, ev_k1_done = nullptr, ev_prep_done = nullptr;
    bool depth_event_set = false, k1_event_set = false, depth_external = false;
    int win_y0 = 0, win_y1 = 0x7fffffff;  // rfx_set_row_window: rows the draws may produce
    int uv_model = RFX_UV_REFERENCE_GL;    // rfx_set_uv_model (the default: the vUv the parity oracle's GL interpolates)
    float2 *coarse = nullptr;  // K1 scratch: exact (min,max) view Z per 16x16 base cell
    unsigned int *cells = nullptr;  // K1 scratch: the march's half-packed (min,max) table
    unsigned int *k1_tiles = nullptr;  // K1 scratch: the persistent march kernel's tile counter
    int n_cu = 0;                      // compute units of the device
    float4 *env = nullptr;     // scene.environment: the whole mip chain, float4 texels
    float *env_marginal = nullptr, *env_conditional = nullptr;  // EquirectHdrInfo inverse-CDF tables (importanceSampling)
    float env_sum_whole = 1.0f, env_sum_decimal = 0.0f;
    int env_w = 0, env_h = 0, env_levels = 0;
    unsigned int env_off[16] = {0};
    Slot slots[RFX_TEX_COUNT];
    // row-tiled runs (rfx_comm.hip): the RCCL communicator of the tile ring, a second stream the exchanges run on, and the two
    // events that order it against the draw stream
    void *comm = nullptr;            // ncclComm_t
    bool comm_owned = false;
    int comm_rank = 0, comm_nranks = 1;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_draws = nullptr, ev_comm = nullptr;
    bool comm_pending = false;       // exchanges issued since the last rfx_comm_wait
    // streaming dumps: a third stream for the host-to-device copies of the NEXT frame and the two events that order it against the draws
    hipStream_t upload_stream = nullptr;
    hipEvent_t ev_staged = nullptr, ev_frame_done = nullptr;
    hipEvent_t ev_batch[2] = {nullptr, nullptr};  // the copies published by the last two flips (recorded on upload_stream)
    unsigned int flips = 0;
    // the peer-load history gather (rfx_peer.hip): this rank's flag block (fine-grained), the device table of every rank's plane and flag block
    // ([0, n) planes, [n, 2 n) flag blocks), what the last call's kernels reported ([0] status bits, [1] texels pulled), the mappings to close
    unsigned long long *peer_flags = nullptr;
    void **peer_table_dev = nullptr;
    unsigned long long *peer_status_dev = nullptr, *peer_status_host = nullptr;
    std::vector<void *> peer_mapped;
    hipEvent_t ev_peer_release = nullptr;  // recorded after the "every rank has pulled" barrier: the next compose draw waits for it
    bool peer_release_pending = false;
    int peer_n = 0, peer_rank = 0, peer_tex = -1;
    unsigned long long peer_epoch = 0;
    // rfx_profile: event pairs around the launches of every draw since the last reset (kind, start, stop), and the events free for re-use
    struct ProfRec { int kind; hipEvent_t a, b; };
    bool profiling = false;
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_free;
    std::string err;
};
void rfx_comm_release(rfx_ctx *c);  // rfx_comm.hip: called by rfx_destroy
void rfx_peer_release(rfx_ctx *c);  // rfx_peer.hip: called by rfx_destroy (before rfx_comm_release: it drains the exchange stream)
// rfx_api.hip, for rfx_comm.hip: enqueue on the draw stream the reduction of the traced rays' history rows into rows_dev[0..1] (min, max)
extern "C" int rfx_internal_hit_rows_enqueue(rfx_ctx *c, int *rows_dev);  // (internal: not part of include/rfx.h)
// ... and of the traced rays' row masks into the first H words of c->hit_mask_dev (allocated here for `ranks` gathered copies)
extern "C" int rfx_internal_hit_mask_enqueue(rfx_ctx *c, int ranks);

extern thread_local std::string g_create_err;

static inline size_t texel_bytes(int id) {
    switch (id) {
    case RFX_TEX_DEPTH: return 4;
    case RFX_TEX_BLUE_NOISE: return 4;
    case RFX_TEX_COMPOSE_RGB: return 12;
    case RFX_TEX_DENOISE_A0: case RFX_TEX_DENOISE_A1: case RFX_TEX_DENOISE_B0: case RFX_TEX_DENOISE_B1: case RFX_TEX_FBCOPY_F16: return 8;
    default: return 16;
    }
}

static inline int fail(rfx_ctx *c, int code, const char *what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess) {
        snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        (void)hipGetLastError();  // reported here: not again by the next launch check of this thread (HIP keeps the last error until it is read)
    } else snprintf(buf, sizeof buf, "%s", what);
    if (c) c->err = buf;
    else g_create_err = buf;
    return code;
}
#define HIPCHK(c, call)                                              \
    do {                                                             \
        hipError_t e__ = (call);                                     \
        if (e__ != hipSuccess) return fail(c, RFX_EDEVICE, #call, e__); \
    } while (0)

