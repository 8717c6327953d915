// kernels.h -- hand-written HIP kernels (gfx950 / CDNA4) for the dpgo RBCD local solve.
//
// Thread mapping ("PC layout"): one lane owns one column c of one pose tile, i.e. the R
// contiguous doubles X[i][c][0..R) of the reference layout (r x (d+1)n column-major,
// include/DPGO/manifold/Poses.h:16-21).  A 64-wide wavefront holds G = 64/(D+1) poses
// (16 for 3-D, 21 for 2-D), a 256-thread workgroup 4G poses.  A wave's loads and stores of
// a dense vector are one contiguous span (G*(D+1)*R*8 bytes).  Operations that couple the
// columns of one pose (tangent projection, Riemannian Hessian correction, block-Jacobi,
// qf retraction) exchange the pose tile through a wave-private LDS slot.
//
// Scalars never leave the device inside a solve: every kernel that produces a dot product
// writes one partial per workgroup; the NEXT kernel's prologue re-reduces those partials in
// every workgroup in a fixed order (bit-identical in all workgroups, deterministic
// run-to-run), advances the tCG / RTR scalar recurrences redundantly in registers, and
// workgroup 0 publishes the new state to the other slot of a two-slot state buffer.
// A kernel boundary (~1.5 us on MI355X) is the cheapest grid-wide barrier on this chip
// (MI355X_MICROARCH.md, price list: barrier-xcd 4-7 us).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dpgo {

#include "kernels/common.h"
#include "kernels/problem.h"
#include "kernels/tcg.h"
#include "kernels/persist.h"
#include "kernels/multilevel.h"
#include "kernels/dense.h"
#include "kernels/manifold.h"
#include "kernels/rtr.h"
#include "kernels/agent.h"
#include "kernels/init.h"

}  // namespace dpgo
