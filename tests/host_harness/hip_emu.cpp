// TEST INFRASTRUCTURE ONLY -- runtime of the CPU stand-in for HIP (see hip/hip_runtime.h in this directory).
#include <hip/hip_runtime.h>

namespace hip_emu {
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local Block* t_block = nullptr;

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
    const unsigned nthreads = block.x * block.y * block.z;
    const unsigned nwaves = (nthreads + 63) / 64;
    Block blk;
    pthread_barrier_init(&blk.all, nullptr, nthreads);
    blk.wave.resize(nwaves);
    blk.slot.assign(nwaves, std::vector<uint64_t>(64, 0));
    for (unsigned w = 0; w < nwaves; ++w) {
        unsigned cnt = std::min(64u, nthreads - w * 64);
        pthread_barrier_init(&blk.wave[w], nullptr, cnt);
    }
    blk.smem.assign(smem_bytes + 64, 0);
    std::vector<std::thread> pool;
    pool.reserve(nthreads);
    for (unsigned t = 0; t < nthreads; ++t) {
        pool.emplace_back([&, t]() {
            t_block = &blk;
            t_blockDim = block;
            t_gridDim = grid;
            t_threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            for (unsigned bz = 0; bz < grid.z; ++bz)
                for (unsigned by = 0; by < grid.y; ++by)
                    for (unsigned bx = 0; bx < grid.x; ++bx) {
                        t_blockIdx = dim3(bx, by, bz);
                        body();
                        pthread_barrier_wait(&blk.all);   // next block only when every thread has finished
                    }
        });
    }
    for (auto& th : pool) th.join();
    pthread_barrier_destroy(&blk.all);
    for (auto& b : blk.wave) pthread_barrier_destroy(&b);
}
}  // namespace hip_emu
