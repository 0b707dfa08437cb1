/*
 * ising_hip_testing.h -- entry points of libising_hip.so that exist for its TESTS only (fault injection, the launch shape).  Not part of the
 * drop-in boundary: a caller that replaces the reference's launch sites needs include/ising_hip.h and nothing from here.
 */
#ifndef ISING_HIP_TESTING_H
#define ISING_HIP_TESTING_H

#include "ising_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* what = 1 leaves the host's record of a slab's completion counters out of step with the device, as a faulted launch would;
 * the next fused launch then gives up after `arg` polls (0: the default bound, ~10 s) and the call that synchronises next
 * returns ISING_E_STATE with tickets and counters reset (tests/test_gpu_fused.py).  what = 2 ages the slab's monotone counters
 * (device and host record together) as billions of sweeps would: the completion counters past the point where the next launch
 * starts them over, the overlapped exchange's counters a few counts before 2^32; results must not change. */
int ising_debug_fault(ising_ctx *ctx, int what, int arg);
/* The same for a batch (ising_batch_*): its completion counters out of step; the next batched launch gives up after `polls`
 * polls and every later batch call reports ISING_E_STATE once, with the batch's tickets and counters reset. */
int ising_batch_debug_fault(ising_batch *b, int polls);
/* The launch shape ising_create picked for this slab's fused launches (tests/test_gpu_policy.py measures it against its neighbours): strip height,
 * workgroups per CU of the persistent grid, and for the split form (ising_sweep_info: 3) the lead (0 otherwise).  Any pointer may be NULL. */
int ising_debug_launch_shape(ising_ctx *ctx, int *strip_rows, int *wg_per_cu, int *split_lead);

#ifdef __cplusplus
}
#endif
#endif /* ISING_HIP_TESTING_H */
