// rfx_comm.hip — the exchanges of a row-tiled run behind the C ABI (include/rfx.h "row-tiled runs"):
// RCCL Send/Recv of halo rows between row neighbours and the all-gather of the composed GI, one process per GPU.
//
// RCCL is bound at run time (dlopen), not at link time: a single-GPU host needs no RCCL at all, and a process that already
// carries one (a torch process maps its own librccl.so.1) must not get a second copy — the already-mapped library is reused.
// The exchanges run on a second stream of the context.  rfx_halo_exchange / rfx_allgather_history order themselves AFTER every
// draw enqueued so far (event on the draw stream) and return; rfx_comm_wait orders every later draw after the exchanges issued
// so far.  Between the two the host may enqueue draws that do not touch the rows in flight — the interior of the tile
// (rfx_set_row_window), the next frame's ray march — which is how the exchange time is hidden.
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include "rfx_ctx.h"

namespace {

// the slice of rccl.h this file uses (ABI of RCCL 2.x: NCCL_UNIQUE_ID_BYTES 128, ncclUint8 == 1, ncclSuccess == 0)
struct NcclUniqueId { char internal[128]; };
typedef void *NcclComm;
typedef int NcclResult;
constexpr int kNcclUint8 = 1, kNcclUint32 = 3;

struct Rccl {
    void *handle = nullptr;
    NcclResult (*GetUniqueId)(NcclUniqueId *) = nullptr;
    NcclResult (*CommInitRank)(NcclComm *, int, NcclUniqueId, int) = nullptr;
    NcclResult (*CommDestroy)(NcclComm) = nullptr;
    NcclResult (*GroupStart)() = nullptr;
    NcclResult (*GroupEnd)() = nullptr;
    NcclResult (*Send)(const void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    NcclResult (*Recv)(void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    NcclResult (*AllGather)(const void *, void *, size_t, int, NcclComm, hipStream_t) = nullptr;
    NcclResult (*Broadcast)(const void *, void *, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(NcclResult) = nullptr;
    std::string why;  // why loading failed
};

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // an RCCL this process already maps (torch's) first, then the ROCm installation's
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char *n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) r.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!r.handle) {
            const char *e = dlerror();  // one call: dlerror() clears the message it returns
            r.why = std::string("librccl.so.1 not loadable: ") + (e ? e : "?");
            return;
        }
#define RFX_SYM(field, name)                                          \
    r.field = (decltype(r.field))dlsym(r.handle, name);              \
    if (!r.field && r.why.empty()) r.why = std::string("RCCL symbol missing: ") + name
        RFX_SYM(GetUniqueId, "ncclGetUniqueId");
        RFX_SYM(CommInitRank, "ncclCommInitRank");
        RFX_SYM(CommDestroy, "ncclCommDestroy");
        RFX_SYM(GroupStart, "ncclGroupStart");
        RFX_SYM(GroupEnd, "ncclGroupEnd");
        RFX_SYM(Send, "ncclSend");
        RFX_SYM(Recv, "ncclRecv");
        RFX_SYM(AllGather, "ncclAllGather");
        RFX_SYM(Broadcast, "ncclBroadcast");
        RFX_SYM(GetErrorString, "ncclGetErrorString");
#undef RFX_SYM
        if (!r.why.empty()) r.handle = nullptr;
    });
    return r.handle ? &r : nullptr;
}

int nccl_fail(rfx_ctx *c, const char *what, NcclResult rc) {
    char buf[384];
    Rccl *r = rccl();
    snprintf(buf, sizeof buf, "%s: %s", what, (r && r->GetErrorString) ? r->GetErrorString(rc) : "RCCL error");
    return fail(c, RFX_EDEVICE, buf);
}
#define NCCLCHK(c, call)                                   \
    do {                                                   \
        NcclResult rc__ = (call);                          \
        if (rc__ != 0) return nccl_fail(c, #call, rc__);   \
    } while (0)

// the exchange stream starts after everything enqueued on the draw stream so far.
// ONE ev_draws / ev_comm pair per context serves every exchange: each call re-records both.  That is correct because the exchange stream is
// in-order — an exchange enqueued later also runs later, so waiting for the LAST recorded ev_comm (rfx_comm_wait) covers every exchange
// issued before it, and a re-recorded ev_draws only ever moves the exchange stream's starting point forward.  (A host that wanted to wait for
// an EARLIER exchange while a later one is still in flight would need one event per exchange; the protocols here never do.)
int comm_begin(rfx_ctx *c) {
    hipSetDevice(c->device);
    hipError_t e = hipEventRecord(c->ev_draws, c->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->comm_stream, c->ev_draws, 0);
    return e == hipSuccess ? RFX_OK : fail(c, RFX_EDEVICE, "rfx_comm: ordering the exchange stream after the draws", e);
}
int comm_end(rfx_ctx *c) {
    hipError_t e = hipEventRecord(c->ev_comm, c->comm_stream);
    if (e != hipSuccess) return fail(c, RFX_EDEVICE, "rfx_comm: hipEventRecord", e);
    c->comm_pending = true;
    return RFX_OK;
}
int ensure_streams(rfx_ctx *c) {
    if (c->comm_stream) return RFX_OK;
    hipSetDevice(c->device);
    hipError_t e = hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_draws, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming);
    return e == hipSuccess ? RFX_OK : fail(c, RFX_EDEVICE, "rfx_comm: stream/event creation", e);
}


// ---- the bounded gather's packed transfer (rfx_gather_history_rows): the history texels a rank needs are column blocks of rows (a bit per
// block in the row's mask word, 32 blocks across the frame).  The owner packs the blocks a peer's mask asks for, row by row and block by
// block, into one contiguous message per peer; the receiver scatters them back.  Both ends derive the layout from the same gathered masks.
__device__ __host__ inline int hist_block_x0(int b, int W) { return (b * W + 31) / 32; }  // first texel of column block b: texel x is in block x * 32 / W
// texel offset of frame column x inside the packed form of a row whose mask is m (x's block bit is set)
__device__ inline int hist_packed_x(unsigned int m, int x, int W) {
    const int b = (x * 32) / W;
    int off = x - hist_block_x0(b, W);
    for (unsigned int below = m & ((1u << b) - 1u); below; below &= below - 1u) {
        const int j = __builtin_ctz(below);
        off += hist_block_x0(j + 1, W) - hist_block_x0(j, W);
    }
    return off;
}
// rows [y0, y1) of the frame-pitched plane `tex` (floats_per_texel floats per texel) <-> the packed staging of one peer.
// row_off[y]: texel offset of row y's packed texels in `staging`, or -1 when nothing of row y travels.
template <bool PACK>
__global__ __launch_bounds__(256) void hist_pack_rows(float *tex, float *staging, const unsigned int *mask, const int *row_off, int W, int y0, int y1, int floats_per_texel) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = y0 + blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= y1) return;
    const unsigned int m = mask[y];
    if (!((m >> ((x * 32) / W)) & 1u) || row_off[y] < 0) return;
    float *t = tex + ((size_t)y * W + x) * floats_per_texel;
    float *q = staging + ((size_t)row_off[y] + hist_packed_x(m, x, W)) * floats_per_texel;
    for (int k = 0; k < floats_per_texel; k++) {
        if (PACK) q[k] = t[k];
        else t[k] = q[k];
    }
}
inline int hist_row_texels(unsigned int m, int W) {  // texels of a row the mask m selects
    int n = 0;
    for (; m; m &= m - 1u) {
        const int j = __builtin_ctz(m);
        n += hist_block_x0(j + 1, W) - hist_block_x0(j, W);
    }
    return n;
}

}  // namespace

void rfx_comm_release(rfx_ctx *c) {
    if (!c) return;
    if (c->comm_stream) hipStreamSynchronize(c->comm_stream);
    if (c->comm && c->comm_owned) {
        Rccl *r = rccl();
        if (r) r->CommDestroy(c->comm);
    }
    c->comm = nullptr;
    c->comm_owned = false;
    if (c->ev_draws) hipEventDestroy(c->ev_draws);
    if (c->ev_comm) hipEventDestroy(c->ev_comm);
    if (c->comm_stream) hipStreamDestroy(c->comm_stream);
    c->ev_draws = c->ev_comm = nullptr;
    c->comm_stream = nullptr;
    c->comm_pending = false;
}

extern "C" {

int rfx_split_rows(int height, int nranks, int rank, int *tile_y0, int *tile_rows) {
    if (height <= 0 || nranks <= 0 || rank < 0 || rank >= nranks) return RFX_EINVAL;
    const int base = (height / nranks) & ~1;  // tile boundaries on even rows
    if (base <= 0) return RFX_EINVAL;
    if (tile_y0) *tile_y0 = rank * base;
    if (tile_rows) *tile_rows = rank == nranks - 1 ? height - rank * base : base;
    return RFX_OK;
}

int rfx_comm_unique_id(void *id128) {
    if (!id128) return RFX_EINVAL;
    Rccl *r = rccl();
    if (!r) return RFX_EUNSUPPORTED;
    NcclUniqueId id;
    if (r->GetUniqueId(&id) != 0) return RFX_EDEVICE;
    memcpy(id128, &id, sizeof id);
    return RFX_OK;
}

int rfx_comm_init(rfx_ctx *c, const void *id128, int rank, int nranks) {
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return RFX_EINVAL;
    Rccl *r = rccl();
    if (!r) return fail(c, RFX_EUNSUPPORTED, "rfx_comm_init: RCCL (librccl.so.1) cannot be loaded on this host");
    if (c->comm) return fail(c, RFX_ESTATE, "rfx_comm_init: this context already has a communicator");
    int y0 = 0, rows = 0;
    if (rfx_split_rows(c->H, nranks, rank, &y0, &rows) != RFX_OK || y0 != c->tile_y0 || rows != c->tile_rows)
        return fail(c, RFX_EINVAL, "rfx_comm_init: the context's tile is not rfx_split_rows(height, nranks, rank)");
    int rc = ensure_streams(c);
    if (rc) return rc;
    hipSetDevice(c->device);
    NcclUniqueId id;
    memcpy(&id, id128, sizeof id);
    NcclComm comm = nullptr;
    NCCLCHK(c, r->CommInitRank(&comm, nranks, id, rank));
    c->comm = comm;
    c->comm_owned = true;
    c->comm_rank = rank;
    c->comm_nranks = nranks;
    return RFX_OK;
}

int rfx_comm_destroy(rfx_ctx *c) {
    if (!c) return RFX_EINVAL;
    hipSetDevice(c->device);
    rfx_comm_release(c);
    return RFX_OK;
}

int rfx_halo_exchange(rfx_ctx *c, rfx_tex id, void *nccl_comm, int up_rank, int down_rank) {
    if (!c || id < 0 || id >= RFX_TEX_COUNT) return RFX_EINVAL;
    Rccl *r = rccl();
    if (!r) return fail(c, RFX_EUNSUPPORTED, "rfx_halo_exchange: RCCL cannot be loaded on this host");
    NcclComm comm = nccl_comm ? nccl_comm : c->comm;
    if (!comm) return fail(c, RFX_ESTATE, "rfx_halo_exchange: no communicator (rfx_comm_init, or pass one)");
    if (c->halo == 0 || (up_rank < 0 && down_rank < 0)) return RFX_OK;
    int rc = ensure_streams(c);
    if (rc) return rc;
    char *base = (char *)rfx_tex_device_ptr(c, id);
    if (!base) return RFX_ENOMEM;
    const Slot &s = c->slots[id];
    const size_t pitch = (size_t)s.width * s.texel;
    const int h = c->halo, lo = c->tile_y0 - s.row0, hi = lo + c->tile_rows;  // tile rows inside the held band
    const int y0 = c->tile_y0, y1 = y0 + c->tile_rows;
    // the rows of the band around the tile that other tiles own must be held
    const int bl = (down_rank >= 0 && y0 - h > 0) ? y0 - h : (down_rank >= 0 ? 0 : y0), bh = (up_rank >= 0 && y1 + h < c->H) ? y1 + h : (up_rank >= 0 ? c->H : y1);
    if (bl < s.row0 || bh > s.row0 + s.rows)
        return fail(c, RFX_EINVAL, "rfx_halo_exchange: the held band does not contain halo_rows rows around the tile (whole-frame slot?)");
    // halo_rows up to the neighbours' height: each neighbour's boundary rows, one Send/Recv pair per direction.  A taller halo reaches past
    // the neighbour (N = 8 at 8K under a fast camera): then every tile whose rows fall inside this tile's band sends them directly, and
    // this tile sends its rows to every tile whose band they fall into — both ends derive the same row intervals from rfx_split_rows, so
    // every Send has its Recv.  That needs rank and size (rfx_comm_init) and the split's neighbours.
    // (the decision must be the same on every rank: the split's smallest tile — tile 0 — against halo_rows, not this rank's neighbourhood)
    const bool known = c->comm && c->comm_nranks > 0;
    int smallest = c->tile_rows;
    if (known) rfx_split_rows(c->H, c->comm_nranks, 0, nullptr, &smallest);
    const bool multi_hop = h > smallest;
    if (multi_hop) {
        if (!known) return fail(c, RFX_EINVAL, "rfx_halo_exchange: halo_rows taller than a tile needs rfx_comm_init (rank and size) on this context");
        if ((up_rank >= 0 && up_rank != c->comm_rank + 1) || (down_rank >= 0 && down_rank != c->comm_rank - 1))
            return fail(c, RFX_EINVAL, "rfx_halo_exchange: halo_rows taller than a tile: up / down must be the split's neighbours (rank + 1 / rank - 1) or -1");
    }
    if ((rc = comm_begin(c))) return rc;
    const size_t bytes = (size_t)h * pitch;
    NCCLCHK(c, r->GroupStart());
    NcclResult e = 0;
    if (multi_hop) {
        for (int p = 0; p < c->comm_nranks && !e; p++) {
            if (p == c->comm_rank || (p > c->comm_rank && up_rank < 0) || (p < c->comm_rank && down_rank < 0)) continue;
            int py0 = 0, pn = 0;
            rfx_split_rows(c->H, c->comm_nranks, p, &py0, &pn);
            const int py1 = py0 + pn;
            int a = y0 > py0 - h ? y0 : py0 - h, b = y1 < py1 + h ? y1 : py1 + h;  // my rows inside p's band
            if (b > a) e = r->Send(base + (size_t)(a - s.row0) * pitch, (size_t)(b - a) * pitch, kNcclUint8, p, comm, c->comm_stream);
            a = py0 > y0 - h ? py0 : y0 - h, b = py1 < y1 + h ? py1 : y1 + h;          // p's rows inside my band
            if (b > a && !e) e = r->Recv(base + (size_t)(a - s.row0) * pitch, (size_t)(b - a) * pitch, kNcclUint8, p, comm, c->comm_stream);
        }
    } else {
        if (up_rank >= 0) {  // `up` owns the rows above this tile: it needs our top rows, we need its bottom rows
            if (!e) e = r->Send(base + (size_t)(hi - h) * pitch, bytes, kNcclUint8, up_rank, comm, c->comm_stream);
            if (!e) e = r->Recv(base + (size_t)hi * pitch, bytes, kNcclUint8, up_rank, comm, c->comm_stream);
        }
        if (down_rank >= 0) {
            if (!e) e = r->Send(base + (size_t)lo * pitch, bytes, kNcclUint8, down_rank, comm, c->comm_stream);
            if (!e) e = r->Recv(base + (size_t)(lo - h) * pitch, bytes, kNcclUint8, down_rank, comm, c->comm_stream);
        }
    }
    NcclResult e2 = r->GroupEnd();
    if (e) return nccl_fail(c, "rfx_halo_exchange: ncclSend/ncclRecv", e);
    if (e2) return nccl_fail(c, "rfx_halo_exchange: ncclGroupEnd", e2);
    return comm_end(c);
}

int rfx_allgather_history(rfx_ctx *c, rfx_tex id, void *nccl_comm) {
    if (!c) return RFX_EINVAL;
    if (id != RFX_TEX_COMPOSE && id != RFX_TEX_COMPOSE_RGB) return fail(c, RFX_EINVAL, "rfx_allgather_history: RFX_TEX_COMPOSE or RFX_TEX_COMPOSE_RGB");
    Rccl *r = rccl();
    if (!r) return fail(c, RFX_EUNSUPPORTED, "rfx_allgather_history: RCCL cannot be loaded on this host");
    NcclComm comm = nccl_comm ? nccl_comm : c->comm;
    if (!comm) return fail(c, RFX_ESTATE, "rfx_allgather_history: no communicator (rfx_comm_init, or pass one)");
    const int n = c->comm_nranks;
    if (nccl_comm && !c->comm) return fail(c, RFX_ESTATE, "rfx_allgather_history: rank and size come from rfx_comm_init");
    int rc = ensure_streams(c);
    if (rc) return rc;
    char *base = (char *)rfx_tex_device_ptr(c, id);  // held whole: frame row y at y * pitch
    if (!base) return RFX_ENOMEM;
    const Slot &s = c->slots[id];
    const size_t pitch = (size_t)s.width * s.texel;
    if ((rc = comm_begin(c))) return rc;
    int y0 = 0, rows = 0, last_rows = 0;
    rfx_split_rows(c->H, n, 0, &y0, &rows);
    rfx_split_rows(c->H, n, n - 1, nullptr, &last_rows);
    if (last_rows == rows) {  // equal tiles: one in-place all-gather (every rank's tile already sits at its frame position)
        NCCLCHK(c, r->AllGather(base + (size_t)c->tile_y0 * pitch, base, (size_t)rows * pitch, kNcclUint8, comm, c->comm_stream));
    } else {  // ragged last tile: one broadcast per owner, aggregated in a group
        NCCLCHK(c, r->GroupStart());
        NcclResult e = 0;
        for (int k = 0; k < n && !e; k++) {
            int ky0 = 0, krows = 0;
            rfx_split_rows(c->H, n, k, &ky0, &krows);
            char *p = base + (size_t)ky0 * pitch;
            e = r->Broadcast(p, p, (size_t)krows * pitch, kNcclUint8, k, comm, c->comm_stream);
        }
        NcclResult e2 = r->GroupEnd();
        if (e) return nccl_fail(c, "rfx_allgather_history: ncclBroadcast", e);
        if (e2) return nccl_fail(c, "rfx_allgather_history: ncclGroupEnd", e2);
    }
    return comm_end(c);
}

int rfx_gather_history_rows(rfx_ctx *c, rfx_tex id, void *nccl_comm, size_t *bytes_received) {
    if (!c) return RFX_EINVAL;
    if (bytes_received) *bytes_received = 0;
    if (id != RFX_TEX_COMPOSE && id != RFX_TEX_COMPOSE_RGB) return fail(c, RFX_EINVAL, "rfx_gather_history_rows: RFX_TEX_COMPOSE or RFX_TEX_COMPOSE_RGB");
    Rccl *r = rccl();
    if (!r) return fail(c, RFX_EUNSUPPORTED, "rfx_gather_history_rows: RCCL cannot be loaded on this host");
    NcclComm comm = nccl_comm ? nccl_comm : c->comm;
    if (!comm) return fail(c, RFX_ESTATE, "rfx_gather_history_rows: no communicator (rfx_comm_init, or pass one)");
    if (nccl_comm && !c->comm) return fail(c, RFX_ESTATE, "rfx_gather_history_rows: rank and size come from rfx_comm_init");
    const int n = c->comm_nranks, me = c->comm_rank;
    int rc = ensure_streams(c);
    if (rc) return rc;
    hipSetDevice(c->device);
    char *base = (char *)rfx_tex_device_ptr(c, id);  // held whole: frame row y at y * pitch
    if (!base) return RFX_ENOMEM;
    const Slot &s = c->slots[id];
    const int H = c->H;
    if (n > 64) return fail(c, RFX_EUNSUPPORTED, "rfx_gather_history_rows: more than 64 ranks");
    // 1. this tile's row mask (device reduction over the trace's hand-over plane: one word per frame row, a bit per column block), on the draw stream
    if ((rc = rfx_internal_hit_mask_enqueue(c, n))) return rc;
    // 2. every rank's mask -> host.  The one host-side wait of the exchange: the plan below needs them (H words per rank: 8.6 KB at 4K).
    if ((rc = comm_begin(c))) return rc;
    NCCLCHK(c, r->AllGather(c->hit_mask_dev, c->hit_mask_dev + H, (size_t)H, kNcclUint32, comm, c->comm_stream));
    HIPCHK(c, hipMemcpyAsync(c->hit_mask_host, c->hit_mask_dev + H, sizeof(unsigned int) * (size_t)n * H, hipMemcpyDeviceToHost, c->comm_stream));
    HIPCHK(c, hipStreamSynchronize(c->comm_stream));
    // 3. rank p needs the column blocks its mask names; whoever owns their rows packs them into ONE message for p (the owner's rows of last
    //    frame's composed GI are current: K4 wrote them), p scatters them back.  Both ends walk the same masks in the same order (row by
    //    row, block by block), so the two sides of every message agree on its size and layout.  Measured on the synthetic orbit
    //    (tools/history_rows_report.py): the blocks are a quarter of the bytes of the rows they lie in — reflections reach most ROWS
    //    below the horizon but only part of each.  (Round 3's plan was the (min, max) row interval per rank.)
    const int W = c->W, fpt = (int)(s.texel / sizeof(float));
    const unsigned int *mine = c->hit_mask_host + (size_t)me * H;
    // row offsets (texels) into the per-peer segments of the two stagings; segment bases per peer
    int *off_host = (int *)(c->hit_mask_host + (size_t)n * H);          // [0, n H): send offsets per peer; [n H, (n + 1) H): receive offsets
    size_t send_base[65], recv_base[65], send_tex = 0, recv_tex = 0;  // [p]: first texel of peer p's segment, [n]: the total
    for (int p = 0; p < n; p++) {
        int py0 = 0, prows = 0;
        rfx_split_rows(H, n, p, &py0, &prows);
        const unsigned int *theirs = c->hit_mask_host + (size_t)p * H;
        send_base[p] = send_tex;
        recv_base[p] = recv_tex;
        int *so = off_host + (size_t)p * H;
        for (int y = 0; y < H; y++) so[y] = -1;
        if (p == me) continue;
        size_t k = 0;
        for (int y = c->tile_y0; y < c->tile_y0 + c->tile_rows; y++)  // what p needs of MY rows
            if (theirs[y]) { so[y] = (int)k; k += (size_t)hist_row_texels(theirs[y], W); }
        send_tex += k;
        k = 0;
        int *ro = off_host + (size_t)n * H;
        for (int y = py0; y < py0 + prows; y++) {  // what I need of p's rows
            ro[y] = -1;
            if (mine[y]) { ro[y] = (int)k; k += (size_t)hist_row_texels(mine[y], W); }
        }
        recv_tex += k;
    }
    for (int y = c->tile_y0; y < c->tile_y0 + c->tile_rows; y++) off_host[(size_t)n * H + y] = -1;  // (my own rows: nothing to receive)
    send_base[n] = send_tex;
    recv_base[n] = recv_tex;
    const size_t need = (send_tex + recv_tex) * s.texel;
    if (need > c->hist_staging_bytes) {
        if (c->hist_staging) { HIPCHK(c, hipStreamSynchronize(c->comm_stream)); hipFree(c->hist_staging); c->hist_staging = nullptr; c->hist_staging_bytes = 0; }
        const size_t cap = need + need / 4 + 4096;
        hipError_t he = hipMalloc((void **)&c->hist_staging, cap);
        if (he != hipSuccess) return fail(c, RFX_ENOMEM, "rfx_gather_history_rows: staging", he);
        c->hist_staging_bytes = cap;
    }
    char *send_stage = (char *)c->hist_staging, *recv_stage = send_stage + send_tex * s.texel;
    int *off_dev = (int *)(c->hit_mask_dev + (size_t)(n + 1) * H);
    HIPCHK(c, hipMemcpyAsync(off_dev, off_host, sizeof(int) * (size_t)(n + 1) * H, hipMemcpyHostToDevice, c->comm_stream));
    const dim3 blk(64, 4);
    for (int p = 0; p < n; p++) {  // pack: one launch per peer over my tile's rows
        const size_t cnt = send_base[p + 1] - send_base[p];
        if (p == me || cnt == 0) continue;
        hipLaunchKernelGGL(hist_pack_rows<true>, dim3((W + 63) / 64, (c->tile_rows + 3) / 4), blk, 0, c->comm_stream, (float *)base, (float *)(send_stage + send_base[p] * s.texel),
                           (const unsigned int *)(c->hit_mask_dev + (size_t)(1 + p) * H), (const int *)(off_dev + (size_t)p * H), W, c->tile_y0, c->tile_y0 + c->tile_rows, fpt);
    }
    HIPCHK(c, hipGetLastError());
    size_t got = 0;
    NCCLCHK(c, r->GroupStart());
    NcclResult e = 0;
    for (int p = 0; p < n && !e; p++) {
        if (p == me) continue;
        const size_t sb = (send_base[p + 1] - send_base[p]) * s.texel, rb = (recv_base[p + 1] - recv_base[p]) * s.texel;
        if (sb) e = r->Send(send_stage + send_base[p] * s.texel, sb, kNcclUint8, p, comm, c->comm_stream);
        if (rb && !e) e = r->Recv(recv_stage + recv_base[p] * s.texel, rb, kNcclUint8, p, comm, c->comm_stream);
        got += rb;
    }
    NcclResult e2 = r->GroupEnd();
    if (e) return nccl_fail(c, "rfx_gather_history_rows: ncclSend/ncclRecv", e);
    if (e2) return nccl_fail(c, "rfx_gather_history_rows: ncclGroupEnd", e2);
    for (int p = 0; p < n; p++) {  // scatter what arrived: one launch per owner over its rows
        const size_t cnt = recv_base[p + 1] - recv_base[p];
        if (p == me || cnt == 0) continue;
        int py0 = 0, prows = 0;
        rfx_split_rows(H, n, p, &py0, &prows);
        hipLaunchKernelGGL(hist_pack_rows<false>, dim3((W + 63) / 64, (prows + 3) / 4), blk, 0, c->comm_stream, (float *)base, (float *)(recv_stage + recv_base[p] * s.texel),
                           (const unsigned int *)(c->hit_mask_dev + (size_t)(1 + me) * H), (const int *)(off_dev + (size_t)n * H), W, py0, py0 + prows, fpt);
    }
    HIPCHK(c, hipGetLastError());
    if (bytes_received) *bytes_received = got;
    return comm_end(c);
}

int rfx_comm_wait(rfx_ctx *c) {
    if (!c) return RFX_EINVAL;
    if (!c->comm_pending) return RFX_OK;
    hipSetDevice(c->device);
    hipError_t e = hipStreamWaitEvent(c->stream, c->ev_comm, 0);
    if (e != hipSuccess) return fail(c, RFX_EDEVICE, "rfx_comm_wait: hipStreamWaitEvent", e);
    c->comm_pending = false;
    return RFX_OK;
}

}  // extern "C"
