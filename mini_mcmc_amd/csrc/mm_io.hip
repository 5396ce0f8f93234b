/*
 * mm_io.hip -- sample sink: CSV in the reference's exact layout (host only, no device code).
 *
 * io/csv.rs:47-69 `save_csv(&Array3<T>, filename)`: header `chain,observation,dim_0,...`, then one record per
 * (chain, observation) in that order with the values formatted by Rust's `Display` -- the shortest decimal string that
 * round-trips, never in exponent notation, integral floats without a fraction ("42"), "NaN", "inf", "-inf" -- records
 * terminated by '\n' (the csv crate's default).  std::to_chars(..., chars_format::fixed) is the same shortest
 * round-trip rule in fixed notation.
 */
#include "../../include/mmcmc.h"

#include <charconv>
#include <cmath>
#include <cstdio>
#include <string>

namespace {
template <class T> void append_value(std::string &s, T v)
{
    if (std::isnan(v)) {
        s += "NaN";
        return;
    }
    if (std::isinf(v)) {
        s += v < 0 ? "-inf" : "inf";
        return;
    }
    char buf[512];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed);
    s.append(buf, r.ptr);
}
} // namespace

extern "C" int mmcmc_save_csv(const void *sample, int dtype, size_t n_chains, size_t n, size_t dim, const char *filename)
{
    if (!filename || (dtype != MMCMC_F32 && dtype != MMCMC_F64) || (!sample && n_chains * n * dim > 0))
        return MMCMC_ERR_INVALID_ARG;
    std::FILE *f = std::fopen(filename, "wb");
    if (!f)
        return MMCMC_ERR_INVALID_ARG;
    std::string line = "chain,observation";
    for (size_t d = 0; d < dim; ++d)
        line += ",dim_" + std::to_string(d);
    line += '\n';
    bool ok = std::fwrite(line.data(), 1, line.size(), f) == line.size();
    for (size_t c = 0; c < n_chains && ok; ++c) {
        for (size_t t = 0; t < n && ok; ++t) {
            line = std::to_string(c);
            line += ',';
            line += std::to_string(t);
            for (size_t d = 0; d < dim; ++d) {
                line += ',';
                const size_t i = (c * n + t) * dim + d;
                if (dtype == MMCMC_F32)
                    append_value(line, static_cast<const float *>(sample)[i]);
                else
                    append_value(line, static_cast<const double *>(sample)[i]);
            }
            line += '\n';
            ok = std::fwrite(line.data(), 1, line.size(), f) == line.size();
        }
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? MMCMC_OK : MMCMC_ERR_STATE;
}
