/*
 * Development-only entry points of libmi355_hotpath_dev.so (the product library built with
 * -DMI355_DEV_BUILD plus csrc/debug_bw.hip): ablation switches and bandwidth probes used by the
 * scripts under tools/.  None of them exists in libmi355_hotpath.so, which therefore carries no
 * process-global mutable state.  No counterpart in the reference.
 */
#ifndef MI355_HOTPATH_DEV_H_
#define MI355_HOTPATH_DEV_H_

#include "mi355_hotpath.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tuning / ablation switches (development only): key 0 selects the decode GEMM variant
 * (0 = single-launch ring kernel, 1 = two-launch kernel), key 2 overrides the split-K plan, key 4 the weight tiles per workgroup. */
void mi355_debug_set(int key, int value);
/* Read-bandwidth probe (development only): streams `bytes` from src with `grid` workgroups. */
int mi355_debug_read_bw(const void* src, int64_t bytes, int grid, int unroll, void* sink,
                        void* stream);
/* Development aid: out[wg] = XCD the workgroup ran on, for a 1-D grid of `grid` workgroups. */
int mi355_debug_xcc_map(void* out, int grid, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355_HOTPATH_DEV_H_ */
