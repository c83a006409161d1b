// emu_runtime.cpp -- fiber scheduler of the CPU emulator (TEST INFRASTRUCTURE ONLY).
// See tests/emu/platform.h.
#include <ucontext.h>

#include <vector>

#include "platform.h"
#ifdef _OPENMP
#include <omp.h>
#endif

thread_local emu_uint3 threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;

namespace emu {

enum State { RUNNABLE = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
enum Op { OP_NONE = 0, OP_SYNC = 1, OP_COUNT = 2, OP_BALLOT = 3, OP_SHFL = 4, OP_GATHER = 5 };

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    int state = DONE;
    int op = OP_NONE;
    int pred = 0;
    unsigned long long bits = 0;
    int src = 0;
    unsigned long long result = 0;
    const void* g_src = nullptr;  // OP_GATHER
    void* g_dst = nullptr;
    size_t g_bytes = 0;
    emu_uint3 tid;
};

static const size_t kStack = 256 * 1024;
static const size_t kMaxThreads = 1024;
static const size_t kLds = 160 * 1024;

struct BlockCtx {
    std::vector<Fiber> fibers;
    ucontext_t main_ctx;
    int cur = -1;
    int nthreads = 0;
    const std::function<void()>* body = nullptr;
    unsigned char* lds = nullptr;
    BlockCtx() {
        fibers.resize(kMaxThreads);
        lds = (unsigned char*)aligned_alloc(256, kLds);
    }
    ~BlockCtx() {
        for (auto& f : fibers) free(f.stack);
        free(lds);
    }
};
static thread_local BlockCtx* tl_ctx = nullptr;

static void fiber_entry() {
    BlockCtx* c = tl_ctx;
    (*c->body)();
    Fiber& f = c->fibers[c->cur];
    f.state = DONE;
    swapcontext(&f.ctx, &c->main_ctx);
}

static void yield_wait(int state, int op) {
    BlockCtx* c = tl_ctx;
    Fiber& f = c->fibers[c->cur];
    f.state = state;
    f.op = op;
    swapcontext(&f.ctx, &c->main_ctx);
}

void sync_block() { yield_wait(WAIT_BLOCK, OP_SYNC); }
int sync_count(int pred) {
    BlockCtx* c = tl_ctx;
    c->fibers[c->cur].pred = pred != 0;
    yield_wait(WAIT_BLOCK, OP_COUNT);
    return (int)c->fibers[c->cur].result;
}
unsigned long long ballot(int pred) {
    BlockCtx* c = tl_ctx;
    c->fibers[c->cur].pred = pred != 0;
    yield_wait(WAIT_WAVE, OP_BALLOT);
    return c->fibers[c->cur].result;
}
unsigned long long shfl_bits(unsigned long long bits, int src_lane) {
    BlockCtx* c = tl_ctx;
    Fiber& f = c->fibers[c->cur];
    f.bits = bits;
    f.src = src_lane;
    yield_wait(WAIT_WAVE, OP_SHFL);
    return c->fibers[c->cur].result;
}
void wave_gather(const void* mine, size_t bytes, void* all) {
    BlockCtx* c = tl_ctx;
    Fiber& f = c->fibers[c->cur];
    f.g_src = mine;
    f.g_dst = all;
    f.g_bytes = bytes;
    yield_wait(WAIT_WAVE, OP_GATHER);
}
int lane() { return tl_ctx->cur & 63; }
void* dyn_lds() { return tl_ctx->lds; }

static void die(const char* msg) {
    fprintf(stderr, "[emu] %s\n", msg);
    abort();
}

static void run_block(BlockCtx* c, dim3 grid, dim3 block, unsigned bx, unsigned by, unsigned bz,
                      const std::function<void()>& body) {
    const int n = (int)(block.x * block.y * block.z);
    if ((size_t)n > kMaxThreads) die("block too large");
    c->nthreads = n;
    c->body = &body;
    blockIdx = emu_uint3{bx, by, bz};
    blockDim = block;
    gridDim = grid;
    for (int i = 0; i < n; ++i) {
        Fiber& f = c->fibers[i];
        if (!f.stack) f.stack = (char*)malloc(kStack);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, fiber_entry, 0);
        f.state = RUNNABLE;
        f.op = OP_NONE;
        f.tid = emu_uint3{(unsigned)(i % block.x), (unsigned)((i / block.x) % block.y),
                          (unsigned)(i / (block.x * block.y))};
    }
    int alive = n;
    const int nwaves = (n + 63) / 64;
    while (alive > 0) {
        for (int i = 0; i < n; ++i) {
            Fiber& f = c->fibers[i];
            if (f.state != RUNNABLE) continue;
            c->cur = i;
            threadIdx = f.tid;
            swapcontext(&c->main_ctx, &f.ctx);
            if (f.state == DONE) alive--;
        }
        if (alive == 0) break;
        bool progressed = false;
        // wave collectives
        for (int w = 0; w < nwaves; ++w) {
            int lo = w * 64, hi = lo + 64 < n ? lo + 64 : n;
            int waiting = 0, live = 0, op = OP_NONE;
            for (int i = lo; i < hi; ++i) {
                Fiber& f = c->fibers[i];
                if (f.state == DONE) continue;
                live++;
                if (f.state == WAIT_WAVE) {
                    waiting++;
                    if (op == OP_NONE) op = f.op;
                    else if (op != f.op) die("wave collective mismatch (divergent ballot/shfl)");
                }
            }
            if (live > 0 && waiting == live) {
                if (op == OP_BALLOT) {
                    unsigned long long m = 0;
                    for (int i = lo; i < hi; ++i)
                        if (c->fibers[i].state == WAIT_WAVE && c->fibers[i].pred) m |= 1ull << (i - lo);
                    for (int i = lo; i < hi; ++i)
                        if (c->fibers[i].state == WAIT_WAVE) c->fibers[i].result = m;
                } else if (op == OP_GATHER) {
                    // all live lanes are suspended here: their payloads (on their own stacks) are stable
                    for (int i = lo; i < hi; ++i) {
                        Fiber& f = c->fibers[i];
                        if (f.state != WAIT_WAVE) continue;
                        for (int s = lo; s < lo + 64; ++s) {
                            char* dst = (char*)f.g_dst + (size_t)(s - lo) * f.g_bytes;
                            if (s < hi && c->fibers[s].state == WAIT_WAVE) memcpy(dst, c->fibers[s].g_src, f.g_bytes);
                            else memset(dst, 0, f.g_bytes);
                        }
                    }
                } else if (op == OP_SHFL) {
                    for (int i = lo; i < hi; ++i) {
                        Fiber& f = c->fibers[i];
                        if (f.state != WAIT_WAVE) continue;
                        int s = lo + (f.src & 63);
                        f.result = (s < hi && c->fibers[s].state == WAIT_WAVE) ? c->fibers[s].bits : f.bits;
                    }
                }
                for (int i = lo; i < hi; ++i)
                    if (c->fibers[i].state == WAIT_WAVE) c->fibers[i].state = RUNNABLE;
                progressed = true;
            }
        }
        // block barrier
        {
            int waiting = 0, cnt = 0, op = OP_NONE;
            for (int i = 0; i < n; ++i) {
                Fiber& f = c->fibers[i];
                if (f.state == WAIT_BLOCK) {
                    waiting++;
                    cnt += f.pred && f.op == OP_COUNT;
                    if (op == OP_NONE) op = f.op;
                    else if (op != f.op) die("block barrier kind mismatch");
                }
            }
            if (waiting == alive) {
                for (int i = 0; i < n; ++i) {
                    Fiber& f = c->fibers[i];
                    if (f.state == WAIT_BLOCK) {
                        f.result = (unsigned long long)cnt;
                        f.state = RUNNABLE;
                    }
                }
                progressed = true;
            }
        }
        if (!progressed) die("deadlock: threads wait at different barriers / collectives");
    }
}

void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body) {
    if (lds_bytes > kLds) die("dynamic LDS request exceeds 160 KiB");
    const long long nblocks = (long long)grid.x * grid.y * grid.z;
    if (nblocks == 0 || block.x * block.y * block.z == 0) die("empty launch");
#pragma omp parallel
    {
        if (!tl_ctx) tl_ctx = new BlockCtx();
        BlockCtx* c = tl_ctx;
#pragma omp for schedule(dynamic, 1)
        for (long long b = 0; b < nblocks; ++b) {
            unsigned bx = (unsigned)(b % grid.x);
            unsigned by = (unsigned)((b / grid.x) % grid.y);
            unsigned bz = (unsigned)(b / ((long long)grid.x * grid.y));
            run_block(c, grid, block, bx, by, bz, body);
        }
    }
}

}  // namespace emu
