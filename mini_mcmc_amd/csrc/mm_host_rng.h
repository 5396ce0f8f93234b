/*
 * mm_host_rng.h -- the reference's own initialisation stream, host side (C++).
 *
 * core.rs:394-435 (`init`, `init_det`, `init_with_seed`) draws the chains' starting points from
 * `SmallRng::seed_from_u64(seed)` + `StandardNormal`, i.e. rand 0.9.4 xoshiro256++ seeded through SplitMix64 and
 * rand_distr 0.5.1's 256-layer ziggurat (crates pinned in the reference's Cargo.lock, not vendored).  To hand a
 * user the SAME starting points as `init_det(n, d)` the engine restates that stream here.  It is used for
 * initial states only -- the samplers themselves use the counter-based stream of mm_rng.h.
 * (oracle/rand_compat.c is a second, independent statement of the same algorithms; tests compare the two.)
 */
#ifndef MM_HOST_RNG_H
#define MM_HOST_RNG_H

#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>

namespace mm_host {

class ZigguratNormal {
  public:
    static constexpr int kLayers = 256;
    static constexpr double kR = 3.654152885361008796;
    static constexpr double kV = 0.00492867323399;
    std::array<double, kLayers + 1> x{}, f{};
    ZigguratNormal()
    {
        auto pdf = [](double t) { return std::exp(-t * t / 2.0); };
        x[0] = kV / pdf(kR);
        x[1] = kR;
        for (int i = 2; i < kLayers; ++i)
            x[i] = std::sqrt(-2.0 * std::log(kV / x[i - 1] + pdf(x[i - 1])));
        x[kLayers] = 0.0;
        for (int i = 0; i <= kLayers; ++i)
            f[i] = pdf(x[i]);
    }
    static const ZigguratNormal &get()
    {
        static const ZigguratNormal z;
        return z;
    }
};

class SmallRng {
    uint64_t s_[4];
    static uint64_t rotl(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }
    static double from_bits(uint64_t mantissa, uint64_t biased_exp)
    {
        uint64_t b = mantissa | (biased_exp << 52);
        double d;
        std::memcpy(&d, &b, sizeof d);
        return d;
    }

  public:
    explicit SmallRng(uint64_t seed)
    {
        uint64_t sm = seed;
        for (auto &w : s_) { /* SplitMix64 */
            sm += 0x9e3779b97f4a7c15ULL;
            uint64_t z = sm;
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
            w = z ^ (z >> 31);
        }
    }
    uint64_t next_u64()
    {
        const uint64_t out = rotl(s_[0] + s_[3], 23) + s_[0];
        const uint64_t t = s_[1] << 17;
        s_[2] ^= s_[0];
        s_[3] ^= s_[1];
        s_[1] ^= s_[2];
        s_[0] ^= s_[3];
        s_[2] ^= t;
        s_[3] = rotl(s_[3], 45);
        return out;
    }
    double uniform() { return (double)(next_u64() >> 11) * 0x1.0p-53; }
    double open01() { return from_bits(next_u64() >> 12, 1023) - (1.0 - 0x1.0p-53); }
    double standard_normal()
    {
        const ZigguratNormal &z = ZigguratNormal::get();
        for (;;) {
            const uint64_t bits = next_u64();
            const unsigned layer = (unsigned)(bits & 0xffu);
            const double u = from_bits(bits >> 12, 1024) - 3.0; /* [-1, 1) */
            const double v = u * z.x[layer];
            if (std::fabs(v) < z.x[layer + 1])
                return v;
            if (layer == 0) {
                double a, b;
                do {
                    a = std::log(open01()) / ZigguratNormal::kR;
                    b = std::log(open01());
                } while (-2.0 * b < a * a);
                return u < 0.0 ? a - ZigguratNormal::kR : ZigguratNormal::kR - a;
            }
            if (z.f[layer + 1] + (z.f[layer] - z.f[layer + 1]) * uniform() < std::exp(-v * v / 2.0))
                return v;
        }
    }
};

} // namespace mm_host

#endif /* MM_HOST_RNG_H */
