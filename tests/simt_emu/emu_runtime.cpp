// SIMT emulator runtime -- TEST INFRASTRUCTURE ONLY.  See include/hip/hip_runtime.h.
#include <hip/hip_runtime.h>

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
        while (b.alive > 0) {
            bool progressed = false;
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
