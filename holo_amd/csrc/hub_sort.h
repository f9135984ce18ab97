// hub_sort.h — the two device-wide radix sorts of the graph build's hub mode (graph_build.hip.h), compiled in their own
// translation unit (hub_sort.hip) because they instantiate rocPRIM templates.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace hspf {

// Stable ascending sort of m 64-bit keys on bits [0, end_bit).  tmp == nullptr: only *tmp_bytes is set (bytes of temporary
// storage the call needs).  Returns a hipError_t.
int hub_sort_keys(void *tmp, size_t *tmp_bytes, const uint64_t *keys_in, uint64_t *keys_out, size_t m, unsigned end_bit,
                  hipStream_t s);
// The same with a 32-bit payload per key.
int hub_sort_pairs(void *tmp, size_t *tmp_bytes, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in,
                   uint32_t *vals_out, size_t m, unsigned end_bit, hipStream_t s);

}  // namespace hspf
