// hub_sort.hip — device-wide stable radix sorts (rocPRIM) behind two plain functions; see hub_sort.h.
// Only the graph build's hub mode calls them: a row that lists more than HUB_DEG links (a LAN pseudonode with thousands
// of members) would make the per-link row scans of kb_links / kb_rank quadratic in that row, so such graphs are built
// from sorted keys instead.  Library code on purpose: this is a rare set-up path, not the SPF path.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "hub_sort.h"

namespace hspf {

int hub_sort_keys(void *tmp, size_t *tmp_bytes, const uint64_t *keys_in, uint64_t *keys_out, size_t m, unsigned end_bit,
                  hipStream_t s) {
  return (int)rocprim::radix_sort_keys(tmp, *tmp_bytes, keys_in, keys_out, m, 0u, end_bit, s);
}

int hub_sort_pairs(void *tmp, size_t *tmp_bytes, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in,
                   uint32_t *vals_out, size_t m, unsigned end_bit, hipStream_t s) {
  return (int)rocprim::radix_sort_pairs(tmp, *tmp_bytes, keys_in, keys_out, vals_in, vals_out, m, 0u, end_bit, s);
}

}  // namespace hspf
