/* tests/hostsim/fakerccl.c — TEST INFRASTRUCTURE.  A stand-in for the slice of librccl.so.1 that csrc/rfx_comm.hip binds at run time
 * (ncclGetUniqueId, ncclCommInitRank, ncclGroupStart/End, ncclSend/Recv, ncclAllGather, ncclBroadcast, ncclCommDestroy,
 * ncclGetErrorString), moving the bytes between the local processes of a test over unix sockets, so that the C ABI's exchange entry points
 * (rfx_halo_exchange, rfx_allgather_history, rfx_comm_wait) can be exercised by `pytest --hostsim` together with the kernels.  It proves the
 * row / offset / peer arithmetic of rfx_comm.hip and the protocol of the hosts above it; it says nothing about RCCL or xGMI.
 * Streams are ignored (the simulator's streams are immediate).  Operations between ncclGroupStart and the matching ncclGroupEnd are queued
 * and executed at the end: one writer thread per peer sends that peer's messages in order while the caller receives. */
#define _GNU_SOURCE
#include <errno.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef struct comm { int rank, n; int *fd; char dir[96]; } *ncclComm_t;
typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4 };

typedef struct { int send; void *ptr; size_t bytes; int peer; ncclComm_t comm; } op_t;
static __thread op_t g_ops[4096];
static __thread int g_nops = 0, g_depth = 0;

static size_t type_size(int t) {
    static const size_t s[] = {1, 1, 4, 4, 8, 8, 2, 4, 8, 2};  /* int8 uint8 int32 uint32 int64 uint64 half float double bf16 */
    return t >= 0 && t < 10 ? s[t] : 1;
}
static int write_all(int fd, const char *p, size_t n) {
    while (n) { ssize_t k = write(fd, p, n); if (k < 0) { if (errno == EINTR) continue; return -1; } p += k; n -= (size_t)k; }
    return 0;
}
static int read_all(int fd, char *p, size_t n) {
    while (n) { ssize_t k = read(fd, p, n); if (k <= 0) { if (k < 0 && errno == EINTR) continue; return -1; } p += k; n -= (size_t)k; }
    return 0;
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fakerccl: socket transport error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof id->internal, "fakerccl-%d-%lx%lx", (int)getpid(), (unsigned long)ts.tv_sec, (unsigned long)ts.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int n, ncclUniqueId id, int rank) {
    if (n < 1 || rank < 0 || rank >= n) return ncclInvalidArgument;
    ncclComm_t c = calloc(1, sizeof *c);
    c->rank = rank; c->n = n; c->fd = malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) c->fd[i] = -1;
    snprintf(c->dir, sizeof c->dir, "/tmp/%.64s", id.internal);
    mkdir(c->dir, 0700);
    /* mesh: rank i listens, every rank j > i connects to it and announces itself */
    struct sockaddr_un a;
    memset(&a, 0, sizeof a);
    a.sun_family = AF_UNIX;
    int ls = -1;
    if (rank < n - 1) {
        ls = socket(AF_UNIX, SOCK_STREAM, 0);
        snprintf(a.sun_path, sizeof a.sun_path, "%s/r%d", c->dir, rank);
        unlink(a.sun_path);
        if (bind(ls, (struct sockaddr *)&a, sizeof a) || listen(ls, n)) return ncclSystemError;
    }
    for (int peer = 0; peer < rank; peer++) {
        int s = socket(AF_UNIX, SOCK_STREAM, 0);
        snprintf(a.sun_path, sizeof a.sun_path, "%s/r%d", c->dir, peer);
        int tries = 0;
        while (connect(s, (struct sockaddr *)&a, sizeof a)) { if (++tries > 3000) return ncclSystemError; usleep(10000); }
        int32_t me = rank;
        if (write_all(s, (const char *)&me, 4)) return ncclSystemError;
        c->fd[peer] = s;
    }
    for (int k = rank + 1; k < n; k++) {
        int s = accept(ls, NULL, NULL);
        int32_t who = -1;
        if (s < 0 || read_all(s, (char *)&who, 4) || who <= rank || who >= n) return ncclSystemError;
        c->fd[who] = s;
    }
    if (ls >= 0) close(ls);
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    for (int i = 0; i < c->n; i++) if (c->fd[i] >= 0) close(c->fd[i]);
    char p[160];
    snprintf(p, sizeof p, "%s/r%d", c->dir, c->rank);
    unlink(p);
    rmdir(c->dir);
    free(c->fd); free(c);
    return ncclSuccess;
}

/* the queued operations of the calling thread, handed to the writer threads */
typedef struct { op_t *ops; int nops, peer, failed; } wjob_t;
static void *wjob(void *arg) {
    wjob_t *j = arg;
    for (int i = 0; i < j->nops; i++)
        if (j->ops[i].send && j->ops[i].peer == j->peer && write_all(j->ops[i].comm->fd[j->peer], j->ops[i].ptr, j->ops[i].bytes)) j->failed = 1;
    return NULL;
}
static ncclResult_t run_ops(void) {
    if (!g_nops) return ncclSuccess;
    ncclComm_t c = g_ops[0].comm;
    pthread_t th[256];
    wjob_t jobs[256];
    int nth = 0, failed = 0;
    for (int peer = 0; peer < c->n && nth < 256; peer++) {
        int any = 0;
        for (int i = 0; i < g_nops; i++) any |= g_ops[i].send && g_ops[i].peer == peer;
        if (!any) continue;
        jobs[nth] = (wjob_t){g_ops, g_nops, peer, 0};
        pthread_create(&th[nth], NULL, wjob, &jobs[nth]);
        nth++;
    }
    for (int i = 0; i < g_nops; i++)
        if (!g_ops[i].send && read_all(g_ops[i].comm->fd[g_ops[i].peer], g_ops[i].ptr, g_ops[i].bytes)) failed = 1;
    for (int t = 0; t < nth; t++) { pthread_join(th[t], NULL); failed |= jobs[t].failed; }
    g_nops = 0;
    return failed ? ncclSystemError : ncclSuccess;
}
static ncclResult_t queue(int send, const void *p, size_t bytes, int peer, ncclComm_t c) {
    if (!c || peer < 0 || peer >= c->n || g_nops >= 4096) return ncclInvalidArgument;
    if (peer == c->rank) return ncclInvalidArgument;  /* self-sends are not used by the library */
    g_ops[g_nops++] = (op_t){send, (void *)p, bytes, peer, c};
    return g_depth ? ncclSuccess : run_ops();
}
ncclResult_t ncclGroupStart(void) { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) { return --g_depth > 0 ? ncclSuccess : run_ops(); }
ncclResult_t ncclSend(const void *p, size_t count, int type, int peer, ncclComm_t c, void *stream) { (void)stream; return queue(1, p, count * type_size(type), peer, c); }
ncclResult_t ncclRecv(void *p, size_t count, int type, int peer, ncclComm_t c, void *stream) { (void)stream; return queue(0, p, count * type_size(type), peer, c); }

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, int type, int root, ncclComm_t c, void *stream) {
    (void)stream;
    const size_t bytes = count * type_size(type);
    ncclResult_t rc = ncclSuccess;
    ncclGroupStart();
    if (c->rank == root) {
        if (recv != send) memmove(recv, send, bytes);
        for (int p = 0; p < c->n && rc == ncclSuccess; p++) if (p != root) rc = queue(1, send, bytes, p, c);
    } else rc = queue(0, recv, bytes, root, c);
    ncclResult_t rc2 = ncclGroupEnd();
    return rc ? rc : rc2;
}
ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, int type, ncclComm_t c, void *stream) {
    (void)stream;
    const size_t bytes = count * type_size(type);
    char *dst = recv;
    if ((const char *)send != dst + (size_t)c->rank * bytes) memmove(dst + (size_t)c->rank * bytes, send, bytes);
    ncclResult_t rc = ncclSuccess;
    ncclGroupStart();
    for (int p = 0; p < c->n && rc == ncclSuccess; p++)
        if (p != c->rank) { rc = queue(1, dst + (size_t)c->rank * bytes, bytes, p, c); if (!rc) rc = queue(0, dst + (size_t)p * bytes, bytes, p, c); }
    ncclResult_t rc2 = ncclGroupEnd();
    return rc ? rc : rc2;
}
