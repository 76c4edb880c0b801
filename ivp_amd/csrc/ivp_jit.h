// ivp_jit.h -- run-time compiled right-hand sides (hiprtc): the device-side `impl IVP for T`.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "ivp_kargs.h"

int ivp_jit_compile(int device, const char *ode_source, int n, int n_params, int n_events, unsigned flags, void **handle, std::string *log);
int ivp_jit_n_events(void *handle);
void ivp_jit_free(void *handle);
void ivp_jit_dims(void *handle, int *n, int *n_params);
const char *ivp_jit_last_log(void *handle);   // hiprtc build log / load error of the most recent failure
hipError_t ivp_jit_launch(void *handle, int what, int method, int fp_mode, int full, const IvpKArgs &a,
                          uint32_t lanes, hipStream_t s);
