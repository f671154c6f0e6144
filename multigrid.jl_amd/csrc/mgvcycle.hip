// mgvcycle.hip - C ABI (include/mgvcycle.h) and level schedule of the MI355X multigrid cycle.
//
// Reference behaviour reproduced (JuliaInv/Multigrid.jl v0.8.0):
//   recursiveCycle  src/Multigrid/MGcycle.jl:1-118   (operation order: SURVEY.md 3.2)
//   relax           src/Multigrid/MGcycle.jl:122-136
//   solveCoarsest   src/Multigrid/MGcycle.jl:138-181 (default branch, l.177)
//   solveMG         src/Multigrid/SolveFuncs.jl:3-39
// The hierarchy is resident in HBM; the host only sequences kernel launches on one HIP stream.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // declarations only: librccl is dlopen'ed by mg_dist_* (single-GPU users do not depend on it)

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstddef>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../include/mgvcycle.h"
#include "mg_kernels.hpp"
#include "mg_march27.hpp"
#include "mg_marchr.hpp"
#include "mg_small.hpp"


// One translation unit, eight parts (round 4: the 6 800-line file split by responsibility; the order is the dependency order):
#include "mg_types.inc"      // Options, DevBuf, Csr, Level, mg_hierarchy
#include "mg_launch.inc"     // byte accounting, profiling slots, kernel launchers
#include "mg_ghost.inc"      // ghost-layer form of the sharded cycle: exchange, validity bookkeeping, global norms
#include "mg_schedule.inc"   // FGMRES relaxation, cycle_level, HIP graphs, solve loop
#include "mg_krylov.inc"     // PCG / BiCGSTAB / FGMRES and block variants
#include "mg_formats.inc"    // upload, format builders, scratch
#include "mg_cabi.inc"       // extern "C": the single-GPU API
#include "mg_dist.inc"       // extern "C": the native multi-GPU sequencer
