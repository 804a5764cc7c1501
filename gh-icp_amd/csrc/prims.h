// Device-wide primitives of the front end, hand-written (prims.hip): inclusive scan, select-by-flag, unique over sorted keys.
// All enqueue on ctx->stream and return without synchronising; counts are written to device memory.
#pragma once
#include "ctx.h"

// data[i] = data[0] + ... + data[i], in place (u32, n <= 2^31)
int gh_scan_inclusive_u32(ghicp_ctx* ctx, unsigned* data, long long n);
// out_idx = the positions i (ascending) with flags[i] != 0; their number to d_count[0]   (was hipcub::DeviceSelect::Flagged over an iota)
int gh_select_flagged_iota(ghicp_ctx* ctx, const unsigned char* flags, long long n, int* out_idx, int* d_count);
// out = vals[i] for the positions i (ascending) with flags[i] != 0   (was hipcub::DeviceSelect::Flagged over a value array)
int gh_select_flagged_u32(ghicp_ctx* ctx, const unsigned* vals, const unsigned char* flags, long long n, unsigned* out, int* d_count);
// out = the distinct values of the SORTED array keys, ascending; their number to d_count[0]   (was hipcub::DeviceSelect::Unique)
int gh_unique_sorted_u32(ghicp_ctx* ctx, const unsigned* keys, long long n, unsigned* out, int* d_count);
// Stable radix sort on the key bits [bit_begin, bit_end), ascending; the input arrays are left untouched, the result is in keys_out / vals_out
// (vals_in == nullptr: keys only).  Spare buffer: B_GRID_TMP; table: B_PRIM_TMP.   (was rocprim::radix_sort_pairs / radix_sort_keys / hipcub::DeviceRadixSort)
int gh_radix_sort_u32(ghicp_ctx* ctx, const unsigned* keys_in, unsigned* keys_out, const unsigned* vals_in, unsigned* vals_out, long long n, int bit_begin, int bit_end);
int gh_radix_sort_u64(ghicp_ctx* ctx, const unsigned long long* keys_in, unsigned long long* keys_out, const unsigned* vals_in, unsigned* vals_out, long long n,
                      int bit_begin, int bit_end);
