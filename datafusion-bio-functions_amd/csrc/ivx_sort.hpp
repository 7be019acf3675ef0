// ivx_sort.hpp -- stable LSD radix sort of SoA multi-word records (see ivx_sort.hip).
#pragma once
#include "ivx_internal.hpp"

// sort by bits [lo,hi) of word `word`; lo and hi byte aligned.  Fields are given
// from the LEAST significant sort criterion to the most significant one.
struct ivx_sort_field { int word, lo, hi; };

// a[0..nw) hold the input arrays, b[0..nw) same-sized scratch; *in_b tells where
// the sorted records ended up (1 = in b).  Uses scratch WS_SORTHIST and WS_SCAN*.
// tight: the fields were packed from the value ranges of the data, so every digit varies (skips the pass that looks
// for constant digits).
// pay (nw == 1 only): pay[0] holds one 32-bit payload per record that travels with it (12-byte records), pay[1] is
// same-sized scratch; the payloads end up in pay[*in_b].
// first_hist (tight sorts of one-word records): the caller has left the digit histograms of the FIRST pass -- bits [lo, lo + 8)
// of the first field -- in WS_SORTHIST already, laid out for ivx_sort_geometry1's workgroups (hist[digit * nblk + block]):
// whoever writes the words can count them in the same pass (sort64's pack kernel), and the sort skips that k_hist.
ivx_status ivx_radix_sort(ivx_ctx *ctx, int nw, u64 *const *a, u64 *const *b, u64 n,
                          const ivx_sort_field *fields, int nfields, int *in_b, bool tight = false, u32 *const *pay = nullptr,
                          bool first_hist = false);
// records per workgroup and workgroups of a sort of n one-word records
void ivx_sort_geometry1(u64 n, u64 *chunk, u32 *nblk);
