// TEST INFRASTRUCTURE ONLY -- scheduler of the wave emulator (see hip/hip_runtime.h here).
#include <hip/hip_runtime.h>

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {

State g;

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

static const size_t kStack = 256 * 1024;

void yield()
{
    Lane *l = g.cur;
    emu_switch(&l->sp, g.sched_sp);
}

static void lane_exit()
{
    Lane *l = g.cur;
    l->done = true;
    Wave &w = g.waves[l->tid >> 6];
    w.alive--;
    g.alive--;
    // a lane that leaves may complete a pending barrier / collective for the others: they will
    // notice on their next poll (arrived == alive)
    emu_switch(&l->sp, g.sched_sp);
    abort();
}

static void trampoline()
{
    g.body();
    lane_exit();
}

static void resolve(Op op, unsigned wave_id, uint64_t /*unused*/)
{
    const unsigned base = wave_id * 64, n = (unsigned)g.lanes.size();
    auto alive = [&](unsigned lane) { return base + lane < n && !g.lanes[base + lane].done; };
    uint64_t ballot = 0;
    if (op == OP_BALLOT)
        for (unsigned i = 0; i < 64; i++)
            if (alive(i) && g.lanes[base + i].in) ballot |= 1ull << i;
    uint64_t first = 0;
    if (op == OP_READFIRST)
        for (unsigned i = 0; i < 64; i++)
            if (alive(i)) { first = g.lanes[base + i].in; break; }
    for (unsigned i = 0; i < 64; i++) {
        if (!alive(i)) continue;
        Lane &l = g.lanes[base + i];
        switch (op) {
        case OP_READFIRST: l.out = first; break;
        case OP_DPP_SHR: l.out = (i >= 1 && alive(i - 1)) ? g.lanes[base + i - 1].in : l.out; break;
        case OP_DPP_SHL: l.out = (i + 1 < 64 && alive(i + 1)) ? g.lanes[base + i + 1].in : l.out; break;
        case OP_BALLOT: l.out = ballot; break;
        case OP_SHFL_XOR: { unsigned s = i ^ l.aux; l.out = (s < 64 && alive(s)) ? g.lanes[base + s].in : l.in; break; }
        case OP_SHFL_UP: l.out = (i >= l.aux && alive(i - l.aux)) ? g.lanes[base + i - l.aux].in : l.in; break;
        }
    }
}

uint64_t collective(Op op, uint64_t in, unsigned aux, uint64_t old)
{
    Lane *l = g.cur;
    Wave &w = g.waves[l->tid >> 6];
    l->in = in; l->aux = aux; l->out = old;
    const uint64_t my_gen = w.gen;
    w.arrived++;
    for (;;) {
        if (w.gen != my_gen) break;
        if (w.arrived >= w.alive) {
            resolve(op, l->tid >> 6, 0);
            w.arrived = 0;
            w.gen++;
            break;
        }
        yield();
    }
    return l->out;
}

void barrier()
{
    const uint64_t my_gen = g.bar_gen;
    g.bar_arrived++;
    for (;;) {
        if (g.bar_gen != my_gen) break;
        if (g.bar_arrived >= g.alive) { g.bar_arrived = 0; g.bar_gen++; break; }
        yield();
    }
}

void launch(dim3 grid, dim3 block, const std::function<void()> &body)
{
    const unsigned n = block.x * block.y * block.z;
    gridDim = grid; blockDim = block;
    g.body = body;
    g.lanes.assign(n, Lane());
    for (unsigned t = 0; t < n; t++) g.lanes[t].stack = (char *)malloc(kStack);
    for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
    for (unsigned bx = 0; bx < grid.x; bx++) {
        g.waves.assign((n + 63) / 64, Wave());
        g.bar_arrived = 0; g.alive = n;
        for (unsigned t = 0; t < n; t++) {
            Lane &l = g.lanes[t];
            l.done = false; l.tid = t; l.gen_seen = 0;
            g.waves[t >> 6].alive++;
            uintptr_t top = ((uintptr_t)l.stack + kStack) & ~(uintptr_t)15;
            void **sp = (void **)(top - 64);
            for (int k = 0; k < 6; k++) sp[k] = nullptr;
            sp[6] = (void *)&trampoline;
            l.sp = sp;
        }
        unsigned remaining = n;
        while (remaining) {
            remaining = 0;
            for (unsigned t = 0; t < n; t++) {
                Lane &l = g.lanes[t];
                if (l.done) continue;
                g.cur = &l;
                threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                blockIdx = dim3(bx, by, bz);
                emu_switch(&g.sched_sp, l.sp);
                if (!l.done) remaining++;
            }
        }
    }
    for (unsigned t = 0; t < n; t++) free(g.lanes[t].stack);
    g.lanes.clear();
}

}  // namespace emu
