// HOST code: the random pixel pick of the attack loop (a1) taken off the Python critical path.
// ref: ibrnet/sample_ray.py:12 (`rng = np.random.RandomState(234)`), :149-171 (`rng.choice(H*W, size=(N_rand,), replace=False)`).
//
// The reference draws every PGD iteration's pixels with numpy's LEGACY RandomState.choice(replace=False), which is
// `permutation(pop)[:size]`: a full Fisher-Yates shuffle of arange(pop) from the back, one masked-rejection draw of the
// MT19937 stream per element.  For a 756x1008 image that is 762 047 sequential draws (~6 ms) per iteration while holding
// the interpreter lock -- as long as all the GPU launches of the iteration take to enqueue.  Keeping the attack
// bit-identical to the reference means consuming the stream exactly like that, so this file restates the published
// algorithm (numpy/random: mt19937.c `mt19937_gen`/`mt19937_next`, distributions.c `random_interval`, _mtrand.pyx
// `_shuffle_raw`) on a caller-supplied copy of the generator state.  It runs without the interpreter lock, so the host
// wrapper (nerfool_amd/ibrnet/sample_ray.py) can compute iteration i+1's pick on a helper thread while iteration i is
// being enqueued, and adopt the advanced state only when that pick is consumed.
#include "nf_common.h"

namespace {
constexpr int MT_N = 624, MT_M = 397;

struct Mt {
    uint32_t* key;
    int pos;
    void refill() {
        int i = 0;
        uint32_t y;
        for (; i < MT_N - MT_M; ++i) {
            y = (key[i] & 0x80000000u) | (key[i + 1] & 0x7fffffffu);
            key[i] = key[i + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; i < MT_N - 1; ++i) {
            y = (key[i] & 0x80000000u) | (key[i + 1] & 0x7fffffffu);
            key[i] = key[i + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        y = (key[MT_N - 1] & 0x80000000u) | (key[0] & 0x7fffffffu);
        key[MT_N - 1] = key[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        pos = 0;
    }
    uint32_t next32() {
        if (pos == MT_N) refill();
        uint32_t y = key[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    uint64_t next64() {                               // mt19937_next64: high word first
        const uint64_t hi = next32();
        return (hi << 32) | next32();
    }
    uint64_t interval(uint64_t max) {                 // uniform integer in [0, max] by masked rejection
        if (max == 0) return 0;
        uint64_t mask = max, v;
        mask |= mask >> 1, mask |= mask >> 2, mask |= mask >> 4, mask |= mask >> 8, mask |= mask >> 16, mask |= mask >> 32;
        if (max <= 0xffffffffull) {
            while ((v = (next32() & mask)) > max) {}
        } else {
            while ((v = (next64() & mask)) > max) {}
        }
        return v;
    }
};
}  // namespace

/* out[0..size) = RandomState.choice(pop, size=(size,), replace=False) for the MT19937 state (key[624], *pos), which is
 * advanced in place exactly as numpy advances it.  scratch: pop int64 (the shuffled arange).  Host-only, no GPU work.
 *
 * Round 5: once a PGD step is one hipGraph launch, this draw IS the host's cost per step (3.1 ms of 3.15 on the GPU box for a
 * 756 x 1008 image), so the common case -- pop < 2^31: every draw is one masked 32-bit word -- runs a tighter loop with the same
 * stream consumption: the generator's 624 words are tempered in one vectorisable pass per refill, the rejection mask is carried
 * (it changes 20 times in 762 047 draws) instead of rebuilt per draw, the rejection is branch-free, and the shuffled arange is
 * 32-bit (3 MB instead of 6: it stays in the core's L2). */
extern "C" int nf_legacy_choice(uint32_t* key, int32_t* pos, int64_t pop, int64_t size, int64_t* out, int64_t* scratch) {
    NF_REQUIRE(key && pos && out && scratch, "nf_legacy_choice: null argument");
    NF_REQUIRE(*pos >= 0 && *pos <= MT_N, "nf_legacy_choice: generator position %d outside [0, 624]", (int)*pos);
    NF_REQUIRE(pop >= 1 && size >= 0 && size <= pop, "nf_legacy_choice: cannot take %lld of %lld without replacement",
               (long long)size, (long long)pop);
    Mt g{key, (int)*pos};
    if (pop <= 0x7fffffffll) {
        uint32_t* x = reinterpret_cast<uint32_t*>(scratch);
        for (int64_t i = 0; i < pop; ++i) x[i] = (uint32_t)i;
        uint32_t temp[MT_N];
        int p = g.pos;
        auto temper_all = [&]() {
            for (int k = 0; k < MT_N; ++k) {
                uint32_t y = key[k];
                y ^= y >> 11;
                y ^= (y << 7) & 0x9d2c5680u;
                y ^= (y << 15) & 0xefc60000u;
                y ^= y >> 18;
                temp[k] = y;
            }
        };
        temper_all();           // (words below p are never read before the next refill)
        uint32_t mask = 0;
        for (uint32_t m = (uint32_t)(pop - 1); m; m >>= 1) mask = (mask << 1) | 1u;
        // one generator word per trip; a rejected word (v > i, up to every second one) swaps x[i] with itself and leaves i where it
        // is -- no data-dependent branch (the rejection branch of the textbook loop mispredicts on a third of the draws)
        for (uint32_t i = (uint32_t)(pop - 1); i >= 1;) {
            while ((mask >> 1) >= i) mask >>= 1;          // smallest 2^k - 1 >= i, as random_interval builds it
            if (p == MT_N) {
                g.refill();
                temper_all();
                p = 0;
            }
            if (p + 12 < MT_N) __builtin_prefetch(&x[temp[p + 12] & mask], 1, 1);      // the array is L3-resident, the index known early
            const uint32_t v = temp[p++] & mask;
            const uint32_t ok = v <= i ? 1u : 0u;
            const uint32_t j = ok ? v : i;
            const uint32_t t = x[i];
            x[i] = x[j];
            x[j] = t;
            i -= ok;
        }
        for (int64_t i = 0; i < size; ++i) out[i] = (int64_t)x[i];
        *pos = p;
        return 0;
    }
    for (int64_t i = 0; i < pop; ++i) scratch[i] = i;
    for (int64_t i = pop - 1; i >= 1; --i) {
        const int64_t j = (int64_t)g.interval((uint64_t)i);
        const int64_t t = scratch[i];
        scratch[i] = scratch[j];
        scratch[j] = t;
    }
    for (int64_t i = 0; i < size; ++i) out[i] = scratch[i];
    *pos = g.pos;
    return 0;
}
