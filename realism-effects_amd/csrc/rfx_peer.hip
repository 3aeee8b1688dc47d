// rfx_peer.hip — the composed GI of a row-tiled run moved by the CONSUMER'S OWN KERNEL: peer loads through IPC mappings (SURVEY.md §8e,
// include/rfx.h "row-tiled runs", the third history-gather mode beside rfx_allgather_history and rfx_gather_history_rows).
//
// xGMI lets a kernel load a peer GPU's HBM directly.  Every rank exports its tile's plane of last frame's composed GI once
// (hipIpcGetMemHandle) together with a small block of flags; every rank opens its peers' handles once.  Per frame, between rfx_ssgi_trace
// and rfx_ssgi_shade, ONE call enqueues on the context's exchange stream:
//   1. a flag barrier — "every rank's compose draw of the previous frame has executed": each rank's barrier kernel runs after that draw in
//      stream order, stores this call's epoch into its slot of every peer's flag block (system-scope release) and waits until every peer's
//      slot in its own block has reached the epoch (system-scope acquire; bounded: a peer that never arrives raises a status bit instead of
//      hanging the device);
//   2. the pull — a kernel walks THIS rank's row mask (one word per frame row, a bit per column block: k1_hit_mask, already on the device)
//      and copies exactly the column blocks its rays will read from their owners' planes into its own; how many texels it moved is counted
//      on the device;
//   3. a second flag barrier — "every rank has pulled": the next compose draw, which overwrites the rows, is ordered after it.
// No host wait, no byte counts from the host, no packing: plan and transfer both live on the device.  The masks of the OTHER ranks are not
// needed at all (rfx_gather_history_rows all-gathers them to size its messages).  N processes on ONE device map each other's allocations
// exactly as N devices would: that is how the tests hold this mode bit-identical to the all-gather; only its xGMI rate needs a second GPU.
#include <string.h>
#include <unistd.h>
#include "rfx_ctx.h"

namespace {

constexpr unsigned int PEER_MAGIC = 0x52465850u;  // "RFXP"
constexpr int PEER_MAX = 64;
struct PeerBlob {  // RFX_PEER_BLOB_BYTES
    unsigned int magic, device;
    long long pid;
    unsigned long long data_ptr, flags_ptr;  // the exporter's own addresses: valid for a context of the SAME process
    unsigned long long plane_bytes;
    hipIpcMemHandle_t data, flags;
    char pad[RFX_PEER_BLOB_BYTES - 40 - 2 * sizeof(hipIpcMemHandle_t)];
};
static_assert(sizeof(PeerBlob) == RFX_PEER_BLOB_BYTES, "the blob of rfx_peer_export");

// flags[r]: rank r's block of PEER_MAX 64-bit slots; slot p of it is written by rank p only
__global__ __launch_bounds__(64) void peer_barrier(unsigned long long *const *flags, int me, int n, unsigned long long epoch, unsigned int *status) {
    const int p = threadIdx.x;
    if (p >= n || p == me) return;
    __hip_atomic_store(&flags[p][me], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long *mine = &flags[me][p];
    for (unsigned int it = 0; __hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch; it++) {
        if (it > (1u << 22)) {  // ~2 s of polling: a peer that never issued its call
            atomicOr(status, 1u);
            return;
        }
        __builtin_amdgcn_s_sleep(16);
    }
}

// the column blocks of rows this rank does not own that its row mask names: from the owner's plane into this rank's, same place
__global__ __launch_bounds__(256) void peer_pull(float *mine, float *const *planes, const unsigned int *mask, int W, int H, int tile_base, int n, int me, int fpt,
                                                 unsigned long long *pulled) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    bool take = false;
    if (x < W && y < H) {
        int owner = y / tile_base;  // rfx_split_rows: equal tiles of tile_base rows, the last one takes the rest
        if (owner > n - 1) owner = n - 1;
        if (owner != me && ((mask[y] >> ((x * 32) / W)) & 1u)) {
            take = true;
            const size_t i = ((size_t)y * W + x) * fpt;
            const float *src = planes[owner] + i;
            for (int k = 0; k < fpt; k++) mine[i + k] = src[k];
        }
    }
    const unsigned long long b = __ballot(take);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(pulled, (unsigned long long)__popcll(b));
}

int ensure_peer_streams(rfx_ctx *c) {
    hipSetDevice(c->device);
    if (!c->comm_stream) {
        hipError_t e = hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_draws, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming);
        if (e != hipSuccess) return fail(c, RFX_EDEVICE, "rfx_peer: stream/event creation", e);
    }
    if (!c->ev_peer_release) {
        hipError_t e = hipEventCreateWithFlags(&c->ev_peer_release, hipEventDisableTiming);
        if (e != hipSuccess) return fail(c, RFX_EDEVICE, "rfx_peer: event creation", e);
    }
    return RFX_OK;
}

}  // namespace

void rfx_peer_release(rfx_ctx *c) {  // rfx_destroy / rfx_peer_close
    if (!c) return;
    hipSetDevice(c->device);
    if (c->comm_stream) hipStreamSynchronize(c->comm_stream);
    for (void *p : c->peer_mapped) hipIpcCloseMemHandle(p);
    c->peer_mapped.clear();
    if (c->peer_table_dev) hipFree(c->peer_table_dev);
    if (c->peer_flags) hipFree(c->peer_flags);
    if (c->peer_status_dev) hipFree(c->peer_status_dev);
    if (c->peer_status_host) hipHostFree(c->peer_status_host);
    if (c->ev_peer_release) hipEventDestroy(c->ev_peer_release);
    c->peer_table_dev = nullptr;
    c->peer_flags = nullptr;
    c->peer_status_dev = nullptr;
    c->peer_status_host = nullptr;
    c->ev_peer_release = nullptr;
    c->peer_release_pending = false;
    c->peer_n = 0;
    c->peer_tex = -1;
}

extern "C" {

int rfx_peer_export(rfx_ctx *c, rfx_tex id, void *blob) {
    if (!c || !blob) return RFX_EINVAL;
    if (id != RFX_TEX_COMPOSE && id != RFX_TEX_COMPOSE_RGB) return fail(c, RFX_EINVAL, "rfx_peer_export: RFX_TEX_COMPOSE or RFX_TEX_COMPOSE_RGB");
    hipSetDevice(c->device);
    const Slot &s = c->slots[id];
    if (s.ptr && !s.owned) return fail(c, RFX_ESTATE, "rfx_peer_export: the plane is bound to a caller's buffer (rfx_bind_external): export that allocation yourself");
    char *base = (char *)rfx_tex_device_ptr(c, id);  // held whole: frame row y at y * pitch
    if (!base) return RFX_ENOMEM;
    if (!c->peer_flags) {
        // fine-grained: the flags are written by peers' kernels while this rank's kernel polls them
        hipError_t e = hipExtMallocWithFlags((void **)&c->peer_flags, PEER_MAX * sizeof(unsigned long long), hipDeviceMallocFinegrained);
        if (e != hipSuccess) return fail(c, RFX_ENOMEM, "rfx_peer_export: flag block", e);
        hipMemset(c->peer_flags, 0, PEER_MAX * sizeof(unsigned long long));
    }
    PeerBlob b;
    memset(&b, 0, sizeof b);
    b.magic = PEER_MAGIC;
    b.device = (unsigned int)c->device;
    b.pid = (long long)getpid();
    b.data_ptr = (unsigned long long)(uintptr_t)base;
    b.flags_ptr = (unsigned long long)(uintptr_t)c->peer_flags;
    b.plane_bytes = (unsigned long long)s.rows * s.width * s.texel;
    hipError_t e = hipIpcGetMemHandle(&b.data, base);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&b.flags, c->peer_flags);
    if (e != hipSuccess) return fail(c, RFX_EUNSUPPORTED, "rfx_peer_export: hipIpcGetMemHandle", e);
    memcpy(blob, &b, sizeof b);
    c->peer_tex = id;
    return RFX_OK;
}

int rfx_peer_open(rfx_ctx *c, rfx_tex id, const void *blobs, int rank, int nranks) {
    if (!c || !blobs || nranks < 2 || nranks > PEER_MAX || rank < 0 || rank >= nranks) return RFX_EINVAL;
    if (c->peer_tex != id || !c->peer_flags) return fail(c, RFX_ESTATE, "rfx_peer_open: rfx_peer_export of the same plane comes first");
    if (c->peer_n) return fail(c, RFX_ESTATE, "rfx_peer_open: peers are already open (rfx_peer_close)");
    int y0 = 0, rows = 0;
    if (rfx_split_rows(c->H, nranks, rank, &y0, &rows) != RFX_OK || y0 != c->tile_y0 || rows != c->tile_rows)
        return fail(c, RFX_EINVAL, "rfx_peer_open: the context's tile is not rfx_split_rows(height, nranks, rank)");
    int rc = ensure_peer_streams(c);
    if (rc) return rc;
    const Slot &s = c->slots[id];
    const PeerBlob *pb = (const PeerBlob *)blobs;
    void *table[2 * PEER_MAX];  // [0, n): planes, [n, 2 n): flag blocks
    const long long pid = (long long)getpid();
    for (int p = 0; p < nranks; p++) {
        const PeerBlob &b = pb[p];
        if (b.magic != PEER_MAGIC || b.plane_bytes != (unsigned long long)s.rows * s.width * s.texel) {
            rfx_peer_release(c);
            return fail(c, RFX_EINVAL, "rfx_peer_open: a blob is not rfx_peer_export's of the same plane and frame size");
        }
        if (p == rank) {
            table[p] = s.ptr;
            table[nranks + p] = c->peer_flags;
        } else if (b.pid == pid) {  // another context of this process: its addresses are ours (other device: peer access is the host's to enable)
            table[p] = (void *)(uintptr_t)b.data_ptr;
            table[nranks + p] = (void *)(uintptr_t)b.flags_ptr;
        } else {
            void *d = nullptr, *f = nullptr;
            hipError_t e = hipIpcOpenMemHandle(&d, b.data, hipIpcMemLazyEnablePeerAccess);
            if (e == hipSuccess) { c->peer_mapped.push_back(d); e = hipIpcOpenMemHandle(&f, b.flags, hipIpcMemLazyEnablePeerAccess); }
            if (e == hipSuccess) c->peer_mapped.push_back(f);
            if (e != hipSuccess) {
                rfx_peer_release(c);
                return fail(c, RFX_EDEVICE, "rfx_peer_open: hipIpcOpenMemHandle", e);
            }
            table[p] = d;
            table[nranks + p] = f;
        }
    }
    hipError_t e = hipMalloc((void **)&c->peer_table_dev, sizeof(void *) * 2 * PEER_MAX);
    if (e == hipSuccess) e = hipMalloc((void **)&c->peer_status_dev, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->peer_status_host, 2 * sizeof(unsigned long long), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMemcpy(c->peer_table_dev, table, sizeof(void *) * 2 * nranks, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(c->peer_status_dev, 0, 2 * sizeof(unsigned long long));
    if (e != hipSuccess) {
        rfx_peer_release(c);
        return fail(c, RFX_ENOMEM, "rfx_peer_open: tables", e);
    }
    c->peer_status_host[0] = c->peer_status_host[1] = 0;
    c->peer_n = nranks;
    c->peer_rank = rank;
    c->peer_epoch = 0;
    return RFX_OK;
}

int rfx_peer_close(rfx_ctx *c) {
    if (!c) return RFX_EINVAL;
    rfx_peer_release(c);
    return RFX_OK;
}

int rfx_peer_gather_history(rfx_ctx *c, rfx_tex id, size_t *bytes_pulled_previous_call) {
    if (!c) return RFX_EINVAL;
    if (bytes_pulled_previous_call) *bytes_pulled_previous_call = 0;
    if (!c->peer_n || c->peer_tex != id) return fail(c, RFX_ESTATE, "rfx_peer_gather_history: rfx_peer_export / rfx_peer_open of this plane come first");
    hipSetDevice(c->device);
    const Slot &s = c->slots[id];
    const int n = c->peer_n, me = c->peer_rank, H = c->H, W = c->W, fpt = (int)(s.texel / sizeof(float));
    // what the PREVIOUS call's kernels reported (its copy has long executed): a peer that never arrived, and the texels pulled
    if (c->peer_epoch > 0) {
        if (c->peer_status_host[0] & 1u) return fail(c, RFX_EDEVICE, "rfx_peer_gather_history: a peer did not reach the previous call's barrier (every rank must issue the call once per frame; contexts of one process on ONE device: include/rfx.h on GPU_MAX_HW_QUEUES)");
        if (bytes_pulled_previous_call) *bytes_pulled_previous_call = (size_t)c->peer_status_host[1] * s.texel;
    }
    // this tile's row mask, on the draw stream (a bit per column block of every frame row this tile's rays read)
    int rc = rfx_internal_hit_mask_enqueue(c, 1);
    if (rc) return rc;
    // the exchange stream starts after every draw enqueued so far: last frame's compose draw, this frame's trace, the mask
    hipError_t e = hipEventRecord(c->ev_draws, c->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->comm_stream, c->ev_draws, 0);
    if (e != hipSuccess) return fail(c, RFX_EDEVICE, "rfx_peer_gather_history: ordering the exchange stream after the draws", e);
    unsigned long long *const *flags = (unsigned long long *const *)(c->peer_table_dev + n);
    unsigned int *status = (unsigned int *)c->peer_status_dev;
    unsigned long long *pulled = c->peer_status_dev + 1;
    HIPCHK(c, hipMemsetAsync(pulled, 0, sizeof(unsigned long long), c->comm_stream));
    const unsigned long long composed = ++c->peer_epoch, pulled_everywhere = ++c->peer_epoch;
    hipLaunchKernelGGL(peer_barrier, dim3(1), dim3(64), 0, c->comm_stream, flags, me, n, composed, status);  // every rank has composed
    int base = 0;
    rfx_split_rows(H, n, 0, nullptr, &base);
    hipLaunchKernelGGL(peer_pull, dim3((W + 63) / 64, (H + 3) / 4), dim3(64, 4), 0, c->comm_stream, (float *)s.ptr, (float *const *)c->peer_table_dev, c->hit_mask_dev, W, H, base, n, me, fpt,
                       pulled);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->peer_status_host, c->peer_status_dev, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->comm_stream));
    // the shade (rfx_comm_wait) needs the pull only ...
    HIPCHK(c, hipEventRecord(c->ev_comm, c->comm_stream));
    c->comm_pending = true;
    // ... the next compose draw needs every rank to have pulled (rfx_compose waits for this event)
    hipLaunchKernelGGL(peer_barrier, dim3(1), dim3(64), 0, c->comm_stream, flags, me, n, pulled_everywhere, status);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev_peer_release, c->comm_stream));
    c->peer_release_pending = true;
    return RFX_OK;
}

}  // extern "C"
