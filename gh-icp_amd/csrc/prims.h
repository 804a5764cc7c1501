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
