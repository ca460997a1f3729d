// SIMT emulator runtime -- TEST INFRASTRUCTURE ONLY.  See include/hip/hip_runtime.h.
#include <hip/hip_runtime.h>
#include <cstring>

namespace emu {

asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
)");

static Block g_blk;
Block& blk() { return g_blk; }

// ET_EMU_DMA=late: LDS-DMA lands at the s_waitcnt that retires it (see hip_runtime.h)
// ET_EMU_SCHED=wave[:seed]: between two workgroup barriers the waves run ONE AT A TIME, in a random order that
// changes every interval -- the adversarial schedule for cross-wave LDS hazards (default: round-robin, all waves
// advance together, which hides a missing barrier).  Both can also be set at run time: emu_configure().
static int g_dma_late = [] { const char* e = getenv("ET_EMU_DMA"); return (e && !strcmp(e, "late")) ? 1 : 0; }();
static int g_sched_seed = [] {
    const char* e = getenv("ET_EMU_SCHED");
    if (!e || strncmp(e, "wave", 4)) return -1;
    return e[4] == ':' ? atoi(e + 5) & 0x7fffffff : 1;
}();
bool dma_late() { return g_dma_late != 0; }
static int sched_seed() { return g_sched_seed; }

static constexpr size_t STACK = 256 * 1024;
static std::vector<char*> g_stacks;

void yield_to_sched() {
    Block& b = g_blk;
    emu_switch(&b.cur->sp, b.sched_sp);
}

static void trampoline() {
    Block& b = g_blk;
    b.body();
    Fiber* f = b.cur;
    for (auto& d : b.dmaq[f->lin]) memcpy(d.dst, d.data, d.size);   // s_endpgm waits for outstanding memory operations
    b.dmaq[f->lin].clear();
    f->st = DONE;
    b.alive--;
    Wave& w = b.waves[f->wave];
    w.alive--;
    // a lane that exits may complete a pending rendezvous of the survivors
    if (w.alive > 0 && w.arrived == w.alive) {
        int base = f->wave * 64;
        for (int i = base; i < base + 64 && i < (int)b.fibers.size(); ++i)
            if (b.fibers[i].st == WAIT_WAVE) b.fibers[i].st = RUN;
        w.arrived = 0;
    }
    if (b.alive > 0 && b.arrived_block == b.alive) {
        for (auto& x : b.fibers) if (x.st == WAIT_BLOCK) x.st = RUN;
        b.arrived_block = 0;
    }
    void* dummy;
    emu_switch(&dummy, b.sched_sp);
    abort();
}

void run_grid(dim3 grid, dim3 block, std::function<void()> body) {
    Block& b = g_blk;
    const int nt = block.x * block.y * block.z;
    while ((int)g_stacks.size() < nt) g_stacks.push_back((char*)aligned_alloc(64, STACK));
    b.bdim = block;
    b.gdim = grid;
    b.body = body;
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        b.bid = {bx, by, bz};
        b.fibers.assign(nt, Fiber{});
        b.waves.assign((nt + 63) / 64, Wave{});
        b.dmaq.assign(nt, {});
        b.alive = nt;
        b.arrived_block = 0;
        for (int t = 0; t < nt; ++t) {
            Fiber& f = b.fibers[t];
            f.lin = t; f.lane = t & 63; f.wave = t >> 6; f.st = RUN;
            f.tid = {(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / (block.x * block.y))};
            b.waves[f.wave].alive++;
            uintptr_t top = ((uintptr_t)g_stacks[t] + STACK) & ~(uintptr_t)15;
            void** sp = (void**)top;
            *--sp = nullptr;                 // alignment pad
            *--sp = (void*)&trampoline;      // return address for the first switch
            for (int i = 0; i < 6; ++i) *--sp = nullptr;
            f.sp = sp;
        }
        const int seed = sched_seed();
        const int nw = (nt + 63) / 64;
        std::vector<int> order(nw);
        for (int i = 0; i < nw; ++i) order[i] = i;
        unsigned rng = 0x9e3779b9u * (unsigned)(seed + 1) + bx * 7919u + by * 104729u;
        while (b.alive > 0) {
            bool progressed = false;
            if (seed >= 0) {
                for (int i = nw - 1; i > 0; --i) {           // new random wave order for this interval
                    rng = rng * 1664525u + 1013904223u;
                    std::swap(order[i], order[(rng >> 8) % (unsigned)(i + 1)]);
                }
                for (int wi = 0; wi < nw; ++wi) {
                    const int w0 = order[wi] * 64, w1 = std::min(nt, w0 + 64);
                    for (bool again = true; again;) {        // this wave alone, until it blocks on the workgroup
                        again = false;
                        for (int t = w0; t < w1; ++t) {
                            Fiber& f = b.fibers[t];
                            if (f.st != RUN) continue;
                            again = progressed = true;
                            b.cur = &f;
                            emu_switch(&b.sched_sp, f.sp);
                        }
                    }
                }
            } else
            for (int t = 0; t < nt; ++t) {
                Fiber& f = b.fibers[t];
                if (f.st != RUN) continue;
                progressed = true;
                b.cur = &f;
                emu_switch(&b.sched_sp, f.sp);
            }
            if (!progressed) {
                fprintf(stderr, "simt_emu: deadlock (divergent barrier / cross-lane op) in block (%u,%u,%u)\n", bx, by, bz);
                abort();
            }
        }
    }
}

}  // namespace emu

// test hook: dma_late 0/1, sched_seed < 0 = round-robin, >= 0 = one wave at a time in a seeded random order
extern "C" void emu_configure(int dma_late, int sched_seed) {
    emu::g_dma_late = dma_late;
    emu::g_sched_seed = sched_seed;
}
