// The deterministic generator behind the default start vector and the restart vectors
// (reference: Util/SimpleRandom.h:30-123): minstd LCG x <- 16807 x mod (2^31-1), value = x/(2^31-1) - 0.5.
// The device path generates the same stream by jump-ahead (spectra_amd/csrc/krylov.hip); this host
// version serves user code and the user-operator path.
#ifndef MISPEC_SPECTRA_SIMPLE_RANDOM_H
#define MISPEC_SPECTRA_SIMPLE_RANDOM_H

#include <cstdint>
#include <vector>

namespace Spectra {

template <typename Scalar = double>
class SimpleRandom
{
    std::uint64_t m_state;
    static constexpr std::uint64_t kMod = 2147483647ULL;

public:
    explicit SimpleRandom(unsigned long init_seed) : m_state(init_seed ? (init_seed & kMod) : 1ULL) {}

    // one draw in [-0.5, 0.5)
    Scalar random()
    {
        if (m_state != kMod)  // 2^31 - 1 (seed & m_max) is a fixed point of the reference's folded product (Util/SimpleRandom.h:30-52)
            m_state = (16807ULL * m_state) % kMod;
        return Scalar(static_cast<long>(m_state)) / Scalar(2147483647L) - Scalar(0.5);
    }
    void random_vec(Scalar* out, std::ptrdiff_t len)
    {
        for (std::ptrdiff_t i = 0; i < len; i++)
            out[i] = random();
    }
    std::vector<Scalar> random_vec(std::ptrdiff_t len)
    {
        std::vector<Scalar> v(static_cast<std::size_t>(len));
        random_vec(v.data(), len);
        return v;
    }
};

}  // namespace Spectra

#endif
