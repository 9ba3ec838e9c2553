// TEST INFRASTRUCTURE ONLY -- runtime of the CPU stand-in for HIP (see hip/hip_runtime.h in this directory).
//
// Execution model: the blocks of a launch are dealt round-robin to a few OS worker threads; inside a worker the threads of a
// block are cooperative fibers (own stacks, a hand-written x86-64 context switch) that run until they reach a block barrier, a
// wave collective or the end of the kernel and then pass control along a ring.  A barrier releases when every fiber that is
// still alive has arrived, so a thread that left the kernel early does not block the others (as on the GPU).  Compared with
// one OS thread per GPU thread and pthread barriers this runs the emulated kernels one to two orders of magnitude faster.
#include <hip/hip_runtime.h>

#include <stdlib.h>
#include <sys/mman.h>

#include <array>
#include <condition_variable>
#include <mutex>

extern "C" void nf_emu_switch(void** save_sp, void* load_sp);
// callee-saved registers + the SSE / x87 control words; everything else is dead across a call by the ABI
asm(R"(
    .text
    .globl nf_emu_switch
    .type nf_emu_switch,@function
nf_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size nf_emu_switch, .-nf_emu_switch
)");

namespace hip_emu {
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;

namespace {
constexpr size_t STACK_BYTES = 256 * 1024, GUARD_BYTES = 4096;

struct Fiber {
    void* sp;
    dim3 tid;
    bool done;
};

struct Block {
    unsigned n = 0, alive = 0, cur = 0;
    std::vector<Fiber> f;
    unsigned all_count = 0, all_gen = 0;
    std::vector<unsigned> wave_count, wave_gen, wave_alive;
    std::vector<std::array<uint64_t, 64>> slot;      // per wave, 64 exchange slots
    std::vector<char> smem;
    void* main_sp = nullptr;
    const std::function<void()>* body = nullptr;
    uint64_t progress = 0;                           // barrier arrivals + finished fibers: a waiter that sees none is deadlocked
};

thread_local Block* t_block = nullptr;
thread_local std::vector<char*> t_stacks;            // fiber stacks of this worker, re-used by every block and launch

void ensure_stacks(unsigned n) {
    while (t_stacks.size() < n) {
        char* p = (char*)mmap(nullptr, STACK_BYTES + GUARD_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == (char*)MAP_FAILED) {
            fprintf(stderr, "hip_emu: cannot map a fiber stack\n");
            abort();
        }
        mprotect(p, GUARD_BYTES, PROT_NONE);         // overflow faults instead of corrupting the neighbour
        t_stacks.push_back(p + GUARD_BYTES);
    }
}

inline void switch_to(Block* b, unsigned idx) {
    Fiber* from = &b->f[b->cur];
    b->cur = idx;
    t_threadIdx = b->f[idx].tid;
    nf_emu_switch(&from->sp, b->f[idx].sp);
}

inline unsigned next_in_block(const Block* b, unsigned i) {
    do {
        i = i + 1 == b->n ? 0 : i + 1;
    } while (b->f[i].done);
    return i;
}

inline unsigned next_in_wave(const Block* b, unsigned i) {
    const unsigned lo = i & ~63u, hi = std::min(b->n, lo + 64);
    do {
        i = i + 1 == hi ? lo : i + 1;
    } while (b->f[i].done);
    return i;
}

[[noreturn]] void deadlock(const char* what) {
    fprintf(stderr, "hip_emu: deadlock at a %s barrier (a lane skipped a collective?)\n", what);
    abort();
}

void fiber_entry() {
    Block* b = t_block;
    (*b->body)();
    b = t_block;
    Fiber& me = b->f[b->cur];
    const unsigned w = b->cur >> 6;
    me.done = true;
    b->alive--;
    b->wave_alive[w]--;
    b->progress++;
    // the others may have been waiting for this fiber only
    if (b->all_count && b->all_count == b->alive) {
        b->all_count = 0;
        b->all_gen++;
    }
    if (b->wave_count[w] && b->wave_count[w] == b->wave_alive[w]) {
        b->wave_count[w] = 0;
        b->wave_gen[w]++;
    }
    if (b->alive == 0) {
        void* dummy;
        nf_emu_switch(&dummy, b->main_sp);
    } else {
        switch_to(b, next_in_block(b, b->cur));
    }
    abort();    // a finished fiber is never resumed
}

void* fresh_stack(char* base) {
    // as nf_emu_switch leaves it: [mxcsr | x87 cw][r15 r14 r13 r12 rbx rbp][return address], the return address 16-byte aligned
    // so that fiber_entry starts with the stack alignment of a called function
    uintptr_t top = ((uintptr_t)base + STACK_BYTES - 64) & ~(uintptr_t)15;
    uint64_t* ret = (uint64_t*)top;
    ret[0] = (uint64_t)(uintptr_t)&fiber_entry;
    uint64_t* regs = ret - 6;
    for (int i = 0; i < 6; ++i) regs[i] = 0;
    uint32_t* ctl = (uint32_t*)(regs - 1);
    ctl[0] = 0x1F80;      // mxcsr: default rounding, exceptions masked
    ctl[1] = 0x037F;      // x87 control word
    return ctl;
}

void run_blocks(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body, unsigned worker, unsigned workers) {
    const unsigned nthreads = block.x * block.y * block.z, nwaves = (nthreads + 63) / 64;
    const uint64_t total = (uint64_t)grid.x * grid.y * grid.z;
    Block blk;
    blk.n = nthreads;
    blk.f.resize(nthreads);
    blk.wave_count.assign(nwaves, 0);
    blk.wave_gen.assign(nwaves, 0);
    blk.wave_alive.assign(nwaves, 0);
    blk.slot.resize(nwaves);
    blk.smem.assign(smem_bytes + 64, 0);
    blk.body = &body;
    t_block = &blk;
    t_blockDim = block;
    t_gridDim = grid;
    ensure_stacks(nthreads);
    for (uint64_t lin = worker; lin < total; lin += workers) {
        t_blockIdx = dim3((unsigned)(lin % grid.x), (unsigned)((lin / grid.x) % grid.y), (unsigned)(lin / ((uint64_t)grid.x * grid.y)));
        blk.alive = nthreads;
        blk.all_count = 0;
        for (unsigned w = 0; w < nwaves; ++w) {
            blk.wave_count[w] = 0;
            blk.wave_alive[w] = std::min(64u, nthreads - w * 64);
        }
        for (unsigned t = 0; t < nthreads; ++t) {
            blk.f[t].tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            blk.f[t].done = false;
            blk.f[t].sp = fresh_stack(t_stacks[t]);
        }
        blk.cur = 0;
        t_threadIdx = blk.f[0].tid;
        nf_emu_switch(&blk.main_sp, blk.f[0].sp);     // returns when the last fiber of the block has finished
    }
    t_block = nullptr;
}

unsigned worker_count() {
    static const unsigned n = [] {
        const char* e = getenv("NF_EMU_THREADS");
        unsigned v = e ? (unsigned)atoi(e) : std::thread::hardware_concurrency();
        return std::max(1u, std::min(v, 64u));
    }();
    return n;
}
}  // namespace

void* dynamic_smem() { return t_block->smem.data(); }

uint64_t* wave_slots() { return t_block->slot[t_block->cur >> 6].data(); }

void wave_barrier() {
    Block* b = t_block;
    const unsigned w = b->cur >> 6;
    b->progress++;
    if (++b->wave_count[w] == b->wave_alive[w]) {
        b->wave_count[w] = 0;
        b->wave_gen[w]++;
        return;
    }
    const unsigned gen = b->wave_gen[w];
    uint64_t seen = b->progress;
    int idle = 0;
    while (b->wave_gen[w] == gen) {
        switch_to(b, next_in_wave(b, b->cur));
        if (b->progress != seen) {
            seen = b->progress;
            idle = 0;
        } else if (++idle > 2) {
            deadlock("wave");
        }
    }
}

void block_barrier() {
    Block* b = t_block;
    b->progress++;
    if (++b->all_count == b->alive) {
        b->all_count = 0;
        b->all_gen++;
        return;
    }
    const unsigned gen = b->all_gen;
    uint64_t seen = b->progress;
    int idle = 0;
    while (b->all_gen == gen) {
        switch_to(b, next_in_block(b, b->cur));
        if (b->progress != seen) {
            seen = b->progress;
            idle = 0;
        } else if (++idle > 2) {
            deadlock("block");
        }
    }
}

namespace {
// persistent helper threads (their fiber stacks live as long as the process); the launching thread is worker 0
struct Pool {
    std::mutex m, launch_m;
    std::condition_variable cv_job, cv_done;
    uint64_t gen = 0;
    unsigned pending = 0, active = 0;
    std::function<void(unsigned)> job;
    std::vector<std::thread> threads;

    void serve(unsigned w) {
        uint64_t seen = 0;
        for (;;) {
            std::function<void(unsigned)> fn;
            unsigned act;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_job.wait(lk, [&] { return gen != seen; });
                seen = gen;
                fn = job;
                act = active;
            }
            if (w < act) fn(w);
            {
                std::lock_guard<std::mutex> lk(m);
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }

    void run(unsigned workers, const std::function<void(unsigned)>& fn) {
        std::lock_guard<std::mutex> one(launch_m);
        const unsigned helpers = worker_count() - 1;
        if (threads.size() < helpers)
            for (unsigned w = (unsigned)threads.size() + 1; w <= helpers; ++w) threads.emplace_back([this, w] { serve(w); });
        {
            std::lock_guard<std::mutex> lk(m);
            job = fn;
            active = workers;
            pending = (unsigned)threads.size();
            gen++;
        }
        cv_job.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};
Pool* pool() {
    static Pool* p = new Pool;      // never destroyed: its threads wait for work until the process exits
    return p;
}
}  // namespace

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
    const uint64_t total = (uint64_t)grid.x * grid.y * grid.z;
    if (total == 0) return;
    const unsigned workers = (unsigned)std::min<uint64_t>(worker_count(), total);
    if (workers == 1) {
        run_blocks(grid, block, smem_bytes, body, 0, 1);
        return;
    }
    pool()->run(workers, [&](unsigned w) { run_blocks(grid, block, smem_bytes, body, w, workers); });
}
}  // namespace hip_emu
