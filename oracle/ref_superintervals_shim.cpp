// ref_superintervals_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A thin extern "C" wrapper that lets the tests drive the REFERENCE's own
// vendored C++ interval structure (superintervals.hpp, the C++ twin of the
// Rust file behind Algorithm::SuperIntervals, interval_join.rs:832-845,
// :892-898) as a second, independent oracle for overlap sets and counts.
//
// The header is NOT copied into this repository: oracle/Makefile compiles
// this shim with -I pointing at the header where it lies under
// /root/reference and writes the result to oracle/_ref/ only.
#include "superintervals.hpp"
#include <cstdint>
#include <vector>

extern "C" {

// Per-key maps; search_values(start,end) returns the values (build rows) of
// all closed intervals overlapping [start,end].
uint64_t ref_si_join(const uint32_t* bkey, const int32_t* bs, const int32_t* be, uint64_t nb,
                     const uint32_t* pkey, const int32_t* ps, const int32_t* pe, uint64_t np,
                     uint32_t* out_build, uint32_t* out_probe, uint64_t cap)
{
    uint32_t nkeys = 0;
    for (uint64_t i = 0; i < nb; i++) if (bkey[i] + 1 > nkeys) nkeys = bkey[i] + 1;
    std::vector<si::IntervalMap<int32_t, uint32_t>> maps(nkeys);
    for (uint64_t i = 0; i < nb; i++) maps[bkey[i]].add(bs[i], be[i], (uint32_t)i);
    for (auto& m : maps) m.build();
    std::vector<uint32_t> found;
    uint64_t n = 0;
    for (uint64_t i = 0; i < np; i++) {
        if (pkey[i] >= nkeys) continue;
        found.clear();
        maps[pkey[i]].search_values(ps[i], pe[i], found);
        for (uint32_t v : found) {
            if (n < cap) { out_build[n] = v; out_probe[n] = (uint32_t)i; }
            n++;
        }
    }
    return n;
}

void ref_si_count(const uint32_t* bkey, const int32_t* bs, const int32_t* be, uint64_t nb,
                  const uint32_t* pkey, const int32_t* ps, const int32_t* pe, uint64_t np,
                  int64_t* out)
{
    uint32_t nkeys = 0;
    for (uint64_t i = 0; i < nb; i++) if (bkey[i] + 1 > nkeys) nkeys = bkey[i] + 1;
    std::vector<si::IntervalMap<int32_t, uint32_t>> maps(nkeys);
    for (uint64_t i = 0; i < nb; i++) maps[bkey[i]].add(bs[i], be[i], (uint32_t)i);
    for (auto& m : maps) m.build();
    for (uint64_t i = 0; i < np; i++)
        out[i] = pkey[i] < nkeys ? (int64_t)maps[pkey[i]].count(ps[i], pe[i]) : 0;
}

}
