// rk_launch.h -- host-side entry points into the two kernel builds (strict / fast).
#pragma once
#include <hip/hip_runtime.h>
#include "ivp_kargs.h"

// lanes per trajectory of the large-n kernels (rk_group.h): hiprtc modules of systems with n <= 16 / n <= 32 put 4 / 2
// trajectories into a wavefront, everything larger (and every built-in) uses the whole wavefront
static inline int ivp_group_width(int n) { return n <= 16 ? 16 : (n <= 32 ? 32 : 64); }

enum { IVP_LAUNCH_INIT = 0, IVP_LAUNCH_CHUNK = 1, IVP_LAUNCH_COOP = 2 /* hiprtc modules only */,
       IVP_LAUNCH_SAMPLE = 3 /* flavour 3's second kernel: one lane per noted step (rk_global.h sample_kernel_t; lean builds, DOP853) */,
       IVP_LAUNCH_EVENTS = 4 /* deferred event refinement: one lane per noted step (rk_global.h event_kernel_t; problems with event functions,
                                flavour 1, explicit methods, IvpKArgs.evd_rec != NULL) */ };

// `lanes` = upper bound of trajectories the launch has to cover (grid = ceil(lanes / 64) one-wave blocks).
// `full` = flavour of the kernel: 0 end state only, 1 the whole device DefaultSolOut, 2 log-only (every accepted step recorded,
// nothing else: no interpolant -- rk_core.h so_log_accepted), 3 deferred t_eval sampling (DOP853, built-in problems without
// event functions: the stepping kernel notes the sampled steps, IVP_LAUNCH_SAMPLE evaluates them -- rk_core.h
// so_defer_samples); tables that do not carry flavour 2 / 3 run them as flavour 1 (the host never asks them for 3).
hipError_t ivp_launch_strict(int what, int method, int rhs_id, int full, const IvpKArgs &a, uint32_t lanes, hipStream_t s);
hipError_t ivp_launch_strict_hoist(int what, int method, int rhs_id, int full, const IvpKArgs &a, uint32_t lanes, hipStream_t s);
hipError_t ivp_launch_fast_hoist(int what, int method, int rhs_id, int full, const IvpKArgs &a, uint32_t lanes, hipStream_t s);
hipError_t ivp_launch_fast(int what, int method, int rhs_id, int full, const IvpKArgs &a, uint32_t lanes, hipStream_t s);

// wave-per-trajectory kernels (rk_group.hip): grid = `trajectories` one-wave blocks; RK23 / DOPRI5 / DOP853 / RK4
hipError_t ivp_launch_group_strict(int what, int method, int rhs_id, int full, const IvpKArgs &a, uint32_t trajectories, hipStream_t s);
hipError_t ivp_launch_group_fast(int what, int method, int rhs_id, int full, const IvpKArgs &a, uint32_t trajectories, hipStream_t s);

// lane-cooperative chunk kernels (rk_coop.h): eight lanes per trajectory, grid = ceil(trajectories / 8) waves;
// DOPRI5 / DOP853, built-in right-hand sides without events
hipError_t ivp_launch_coop_strict(int method, int rhs_id, int full, const IvpKArgs &a, uint32_t trajectories, hipStream_t s);
hipError_t ivp_launch_coop_fast(int method, int rhs_id, int full, const IvpKArgs &a, uint32_t trajectories, hipStream_t s);

// thread-per-trajectory BDF kernels (rk_bdf.hip: pinned-coefficient build, n <= 8)
hipError_t ivp_launch_bdf_strict(int what, int rhs_id, int full, const IvpKArgs &a, uint32_t lanes, hipStream_t s);
hipError_t ivp_launch_bdf_fast(int what, int rhs_id, int full, const IvpKArgs &a, uint32_t lanes, hipStream_t s);
// ... and the same source under __launch_bounds__(64, 2) for batches that over-subscribe the chip (see rk_bdf.hip)
hipError_t ivp_launch_bdf_strict_occ2(int what, int rhs_id, int full, const IvpKArgs &a, uint32_t lanes, hipStream_t s);
hipError_t ivp_launch_bdf_fast_occ2(int what, int rhs_id, int full, const IvpKArgs &a, uint32_t lanes, hipStream_t s);
