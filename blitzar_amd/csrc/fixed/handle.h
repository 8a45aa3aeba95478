// `sxt_multiexp_handle`: generators fixed at creation, reused by many multiexponentiations
// (reference: cbnb::multiexp_handle {curve_id, partition_table_accessor},
// sxt/cbindings/base/multiexp_handle.h:28-31).  The native handle keeps one resident addend per
// generator in HBM instead of the reference's 2^w-entry partition tables (64 GiB for 2^18 Grumpkin
// generators at w = 16); the table format only appears at the file boundary
// (fixed/partition_table.h).
#pragma once

#include <vector>

#include "blitzar_amd/csrc/api/state.h"
#include "blitzar_amd/csrc/msm/dispatch.h"

namespace bz {
struct multiexp_handle {
  const curve_vtable* vt = nullptr;
  u64 n = 0;                 // generators (file-loaded handles include the identity padding)
  unsigned window_width = 16; // only used when the handle is written to a file
  std::vector<u8> host_projective; // n projective elements (host copy)
  // GPU backend: resident addends, one replica per device the backend drives (indexed by
  // device_state::slot; 16 MiB for 2^18 Grumpkin generators)
  std::vector<resident_table> tables;
  std::vector<int> devices;
  const resident_table* table_on(int device) const {
    for (size_t k = 0; k < devices.size(); ++k) {
      if (devices[k] == device) return &tables[k];
    }
    return nullptr;
  }
};
} // namespace bz
