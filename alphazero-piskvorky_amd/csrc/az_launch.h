// az_launch.h -- per-board-size kernel launchers.  Every board size n = 3..15 is its own translation unit
// (az_kernels.hip compiled with -DAZ_N=n, in parallel); the engine dispatches through this table.
#pragma once
#include "az_net.h"
#include "az_net_emul.h"
#include "az_search.h"

struct LaunchCtx {
    hipStream_t stream;
    DevState d;         // the games' view: B = game slots
    DevState dv;        // the net kernels' view: B = evaluation items (slots x leaves per batch), s_status / s_net per item;
                        // identical to d unless virtual-loss batching is on
    NetWeights w[2];
    ResWeights rw[2];
    int model;          // AZ_MODEL_PLAIN | AZ_MODEL_RESNET
    int synthetic;      // AZ_EVAL_SYNTHETIC
    int persist_gp;     // games per workgroup of the persistent search kernel, 0 = lock-step pipeline (part of the graph key)
    int vl_kernel;      // the batched (virtual-loss) tree kernel is in use (part of the graph key)
    int emul;           // 0 | AZ_TRUNK_BF16X3 | AZ_TRUNK_F16X2: the convs on the 16-bit MFMA with split operands (az_net_emul.h)
    float *feat;
    unsigned long long *dbg;
    float *scratch;     // split-trunk images, SizeOps::split_floats_per_group floats per board group (or null)
};

struct SizeOps {
    void (*trunk)(const LaunchCtx &, int net_id);
    void (*trunk_split)(const LaunchCtx &, int net_id);     // same results as trunk, lower latency for few boards
    long long (*split_scratch_floats)(int slots, int model);
    void (*fc)(const LaunchCtx &, int net_id);
    void (*step)(const LaunchCtx &, int rootN, int do_select);
    void (*step_vl)(const LaunchCtx &, int sims_done, int nb_next);      // virtual-loss batching (DevState.L leaves per game)
    void (*root_cache)(const LaunchCtx &);                                // evaluation-cache lookup of the root positions
    // persistent search kernel (az_search.h): prepare returns 1 when `games` games per workgroup with S simulations fit
    // into LDS on the current device (and raises the kernel's dynamic-LDS limit), 0 when this size has no such kernel
    int (*search_prepare)(int S, int games, int synthetic, int model);
    void (*search)(const LaunchCtx &, int games);
    void (*move)(const LaunchCtx &);
    void (*eval_tail)(const LaunchCtx &, int count, float *policy, float *value);
};

const SizeOps *az_size_ops(int n);     // nullptr for unsupported sizes
// weak: a development build may leave sizes out (make SIZES="5 9 15"); az_size_ops then returns nullptr for them
#define AZ_DECL_OPS(n) const SizeOps *az_size_ops_##n() __attribute__((weak));
AZ_DECL_OPS(3) AZ_DECL_OPS(4) AZ_DECL_OPS(5) AZ_DECL_OPS(6) AZ_DECL_OPS(7) AZ_DECL_OPS(8) AZ_DECL_OPS(9)
AZ_DECL_OPS(10) AZ_DECL_OPS(11) AZ_DECL_OPS(12) AZ_DECL_OPS(13) AZ_DECL_OPS(14) AZ_DECL_OPS(15)
