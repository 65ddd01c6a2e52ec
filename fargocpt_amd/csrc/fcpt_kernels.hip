// HIP kernels of the gas update for gfx950 (CDNA4).  FP64 throughout, phi is the
// contiguous (coalesced) axis of every grid, no MFMA (there is no contraction).
//
// Each kernel cites the reference loop nest it restates (paths relative to the
// reference's src/).  Operand order follows the reference so results agree with
// the CPU path to rounding.
//
// One translation unit; the kernels live in kernels/*.h by topic (included below, in this order).
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdlib>
#include <cstring>

#include "fcpt_kernels.h"

namespace fcpt {

#include "kernels/device_util.h"
#include "kernels/source_loops.h"
#include "kernels/source_fused.h"
#include "kernels/boundary.h"
#include "kernels/source_march.h"
#include "kernels/transport.h"
#include "kernels/transport_fused.h"
#include "kernels/cfl.h"
#include "kernels/launch.h"

} // namespace fcpt
