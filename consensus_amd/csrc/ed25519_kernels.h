#pragma once
#include <hip/hip_runtime.h>

#include "ed25519_core.h"

namespace sbv {
hipError_t launch_ed25519_verify(const uint8_t* d_tuples, size_t n, u32* d_qtab, const aniels* d_btab, uint8_t* d_bitmap,
                                 hipStream_t stream);
void host_build_ed_btable(aniels* out);   // 32 x 128 affine-Niels multiples of B (one-time setup)
#define SBV_ED_BTAB_ENTRIES (SBV_ED_BTAB_WINDOWS * SBV_ED_BTAB_PER_WINDOW)
}  // namespace sbv
