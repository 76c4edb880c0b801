// log_gather.h -- host entry points of log_gather.hip (one-pass accepted-step log: page chains -> CSR).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// offsets[b] = sum of n_log[0 .. b) for b = 0 .. B (offsets[B] = total number of records); `scratch` holds
// ivp_log_scan_scratch_bytes(B) bytes of device memory
size_t ivp_log_scan_scratch_bytes(size_t B);
hipError_t ivp_log_scan(const uint32_t *n_log, size_t B, unsigned long long *offsets, void *scratch, hipStream_t s);

// Every wave page of the pool (`subs` sub-pools of `region` doubles, `alloc` = their counters, max_arenas = the largest
// directory count among them) to t_log[dst_base + offsets[j] + k], y_log[(dst_base + offsets[j] + k) * n + c].  Nothing is
// written when offsets[B] > capacity.
hipError_t ivp_log_gather(const double *pool, unsigned long long region, const unsigned long long *alloc, uint32_t subs, uint32_t max_arenas,
                          const unsigned long long *offsets, size_t B, int n, unsigned long long capacity, unsigned long long dst_base,
                          double *t_log, double *y_log, hipStream_t s);
