// tests/hostsim/hostsim.cpp — TEST INFRASTRUCTURE (see hip/hip_runtime.h): the simulator's execution state and its block scheduler.
//
// One block = N fibers on the calling OS thread (an OpenMP worker), scheduled cooperatively:
//   * every runnable fiber runs until it blocks (barrier / wave operation) or returns;
//   * then, per wavefront (64 consecutive linear thread ids), a pending wave operation is resolved among the lanes that wait at it —
//     every other lane of that wave is at a barrier or finished, i.e. does not take part, exactly the device's exec-mask semantics;
//   * when no wave operation is pending, every unfinished fiber waits at the barrier, which is released.
// Anything else is a deadlock on the device too (a barrier in divergent control flow) and aborts with a message.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HOSTSIM_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void **fake_stack_save, const void *bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void *fake_stack_save, const void **bottom_old, size_t *size_old);
#endif
#endif

// bench.py asks the loaded library whether it is this simulator (no device to select, drain or profile then)
extern "C" int rfx_hostsim_build(void) { return 1; }
size_t hostsim_last_shmem = 0;
extern "C" unsigned long long rfx_hostsim_last_dynamic_lds(void) { return (unsigned long long)hostsim_last_shmem; }

thread_local hostsim_idx threadIdx, blockIdx, blockDim, gridDim;
thread_local unsigned char *hostsim_lds = nullptr;

// void hostsim_switch(void **save_sp, void *load_sp): push the callee-saved registers, park the stack pointer, adopt the other one
extern "C" void hostsim_switch(void **save_sp, void *load_sp);
asm(R"(
    .text
    .globl hostsim_switch
    .type hostsim_switch,@function
hostsim_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hostsim_switch, .-hostsim_switch
)");

namespace {
enum { RUNNABLE = 0, AT_BARRIER = 1, AT_WAVEOP = 2, DONE = 3 };
#ifdef HOSTSIM_ASAN
constexpr size_t STACK_BYTES = 256u << 10;
#else
constexpr size_t STACK_BYTES = 512u << 10;
#endif
struct Fiber {
    void *sp;
    unsigned char *stack;
    hostsim_idx idx;
    int state, kind, arg;
    unsigned long long payload, result;
    void *site;  // return address of the wave operation the fiber waits at (diagnostics)
    void *fake_stack;
};
thread_local Fiber *fibers = nullptr;
thread_local unsigned int fibers_alloc = 0;
thread_local Fiber *cur = nullptr;
thread_local void *sched_sp = nullptr;
thread_local void *sched_fake = nullptr;
thread_local const void *sched_bottom = nullptr;
thread_local size_t sched_size = 0;
thread_local void (*blk_call)(void *) = nullptr;
thread_local void *blk_ctx = nullptr;

void to_scheduler(bool forever) {  // from the running fiber
    Fiber *f = cur;
#ifdef HOSTSIM_ASAN
    __sanitizer_start_switch_fiber(forever ? nullptr : &f->fake_stack, sched_bottom, sched_size);
#endif
    (void)forever;
    hostsim_switch(&f->sp, sched_sp);
#ifdef HOSTSIM_ASAN
    __sanitizer_finish_switch_fiber(f->fake_stack, &sched_bottom, &sched_size);
#endif
}
void fiber_main() {
#ifdef HOSTSIM_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &sched_bottom, &sched_size);
#endif
    blk_call(blk_ctx);
    cur->state = DONE;
    to_scheduler(true);
    std::abort();
}
void resume(Fiber *f) {  // from the scheduler
    cur = f;
    threadIdx = f->idx;
#ifdef HOSTSIM_ASAN
    __sanitizer_start_switch_fiber(&sched_fake, f->stack, STACK_BYTES);
#endif
    hostsim_switch(&sched_sp, f->sp);
#ifdef HOSTSIM_ASAN
    __sanitizer_finish_switch_fiber(sched_fake, nullptr, nullptr);
#endif
}
void ensure_fibers(unsigned int n) {
    if (n <= fibers_alloc) return;
    Fiber *nf = (Fiber *)std::calloc(n, sizeof(Fiber));
    if (fibers) std::memcpy(nf, fibers, fibers_alloc * sizeof(Fiber));
    for (unsigned int i = fibers_alloc; i < n; i++) {
        void *m = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == MAP_FAILED) { std::fprintf(stderr, "hostsim: cannot map a fiber stack\n"); std::abort(); }
        nf[i].stack = (unsigned char *)m;
    }
    std::free(fibers);
    fibers = nf;
    fibers_alloc = n;
}
}  // namespace

void hostsim_barrier_wait() {
    cur->state = AT_BARRIER;
    to_scheduler(false);
}
unsigned long long hostsim_wave_exchange(int kind, unsigned long long payload, int arg) {
    cur->site = __builtin_return_address(0);
    Fiber *f = cur;
    f->state = AT_WAVEOP;
    f->kind = kind;
    f->arg = arg;
    f->payload = payload;
    to_scheduler(false);
    return f->result;
}

void hostsim_run_block(unsigned int n, unsigned int bx, unsigned int by, void (*call)(void *), void *ctx) {
    ensure_fibers(n);
    blk_call = call;
    blk_ctx = ctx;
    for (unsigned int i = 0; i < n; i++) {
        Fiber &f = fibers[i];
        f.idx = {i % bx, (i / bx) % by, i / (bx * by)};
        f.state = RUNNABLE;
        f.fake_stack = nullptr;
        // initial frame: six callee-saved registers, the entry point as the return address of hostsim_switch, and one pad word so that
        // fiber_main starts with the stack alignment of a called function (rsp = 8 mod 16)
        void **top = (void **)(f.stack + STACK_BYTES);
        void **base = top - 8;
        for (int k = 0; k < 8; k++) base[k] = nullptr;
        base[6] = (void *)&fiber_main;
        f.sp = (void *)base;
    }
    for (;;) {
        for (unsigned int i = 0; i < n; i++)
            if (fibers[i].state == RUNNABLE) resume(&fibers[i]);
        // nobody is runnable now
        unsigned int done = 0, at_barrier = 0;
        bool released = false;
        for (unsigned int w0 = 0; w0 < n; w0 += 64) {
            const unsigned int w1 = w0 + 64 < n ? w0 + 64 : n;
            unsigned long long here = 0;
            int kind = 0;
            // HOSTSIM_JOIN is the reconvergence point of a divergent region (lanes that left the region early wait there for the others, as
            // the device's exec mask makes them): it completes only when EVERY unfinished lane of the wavefront has arrived; while lanes are
            // still inside the region, their wave operations complete among themselves and the joiners keep waiting.
            unsigned int joiners = 0, unfinished = 0;
            for (unsigned int i = w0; i < w1; i++) {
                unfinished += fibers[i].state != DONE;
                joiners += fibers[i].state == AT_WAVEOP && fibers[i].kind == HOSTSIM_JOIN;
            }
            const bool join_now = joiners && joiners == unfinished;
            for (unsigned int i = w0; i < w1; i++)
                if (fibers[i].state == AT_WAVEOP && (fibers[i].kind != HOSTSIM_JOIN || join_now)) {
                    here |= 1ull << (i - w0);
                    if (kind && kind != fibers[i].kind) {
                        std::fprintf(stderr, "hostsim: the lanes of a wavefront wait at different wave operations (kinds per lane of the wavefront, . = not at one:");
                        for (unsigned int j = w0; j < w1; j++) std::fprintf(stderr, " %c", fibers[j].state == AT_WAVEOP ? (char)('0' + fibers[j].kind) : fibers[j].state == DONE ? 'x' : '.');
                        std::fprintf(stderr, ")\nsites (addr2line -f -i -e <lib> <offset>):");
                        for (unsigned int j = w0; j < w1; j++) if (fibers[j].state == AT_WAVEOP) std::fprintf(stderr, " %u:%d:%p", j - w0, fibers[j].kind, fibers[j].site);
                        std::fprintf(stderr, "\n");
                        std::abort();
                    }
                    kind = fibers[i].kind;
                }
            if (!here) continue;
            unsigned long long ballot = 0;
            int first = -1;
            for (unsigned int i = w0; i < w1; i++)
                if (here >> (i - w0) & 1) {
                    if (first < 0) first = (int)i;
                    if (fibers[i].payload & 1) ballot |= 1ull << (i - w0);
                }
            for (unsigned int i = w0; i < w1; i++) {
                if (!(here >> (i - w0) & 1)) continue;
                Fiber &f = fibers[i];
                switch (kind) {
                case HOSTSIM_BALLOT: f.result = ballot; break;
                case HOSTSIM_READFIRST: f.result = fibers[first].payload; break;
                case HOSTSIM_SHFL_XOR:
                case HOSTSIM_SHFL: {
                    const unsigned int lane = i - w0, pl = kind == HOSTSIM_SHFL_XOR ? (lane ^ (unsigned int)f.arg) & 63u : (unsigned int)f.arg & 63u;
                    f.result = (w0 + pl < w1 && (here >> pl & 1)) ? fibers[w0 + pl].payload : f.payload;
                    break;
                }
                case HOSTSIM_JOIN: f.result = 0ull; break;
                case HOSTSIM_BPERMUTE: {
                    const unsigned int pl = (unsigned int)f.arg & 63u;
                    f.result = (w0 + pl < w1 && (here >> pl & 1)) ? fibers[w0 + pl].payload : 0ull;
                    break;
                }
                case HOSTSIM_PERMUTE: {
                    f.result = 0ull;
                    for (unsigned int j = w0; j < w1; j++)
                        if ((here >> (j - w0) & 1) && ((unsigned int)fibers[j].arg & 63u) == i - w0) f.result = fibers[j].payload;
                    break;
                }
                default: std::fprintf(stderr, "hostsim: unknown wave operation %d\n", kind); std::abort();
                }
            }
            for (unsigned int i = w0; i < w1; i++)
                if (here >> (i - w0) & 1) fibers[i].state = RUNNABLE;
            released = true;
        }
        if (released) continue;
        for (unsigned int i = 0; i < n; i++) {
            done += fibers[i].state == DONE;
            at_barrier += fibers[i].state == AT_BARRIER;
        }
        if (done == n) break;
        if (done + at_barrier != n) { std::fprintf(stderr, "hostsim: deadlock (%u of %u threads finished, %u at the barrier)\n", done, n, at_barrier); std::abort(); }
        for (unsigned int i = 0; i < n; i++)
            if (fibers[i].state == AT_BARRIER) fibers[i].state = RUNNABLE;
    }
}
