// az_kernels.hip -- the size-templated kernels instantiated for ONE board size (compile with -DAZ_N=n).
#include <hip/hip_runtime.h>
#include "az_launch.h"

#ifndef AZ_N
#error "compile with -DAZ_N=<board size>"
#endif
#define AZ_CAT2(a, b) a##b
#define AZ_CAT(a, b) AZ_CAT2(a, b)

namespace {
constexpr int N = AZ_N;

void trunk(const LaunchCtx &c, int net_id)
{
    if (c.model == 1) {
        typedef ResGeo<N> G;
        dim3 gt((c.dv.B + G::G - 1) / G::G), bt(G::NW * 64);
        hipLaunchKernelGGL(k_trunk_res<N>, gt, bt, 0, c.stream, c.dv, c.rw[net_id], net_id, c.feat);
    } else {
        typedef NetGeo<N> G;
        const int ngroups = (c.dv.B + G::G - 1) / G::G;
        dim3 gt(AZ_SEQ == 0 ? (ngroups < 256 ? ngroups : 256) : (ngroups + AZ_SEQ - 1) / AZ_SEQ), bt(G::NW * 64);
        hipLaunchKernelGGL(k_trunk<N>, gt, bt, 0, c.stream, c.dv, c.w[net_id], net_id, c.feat, c.dbg);
    }
}

void trunk_split(const LaunchCtx &c, int net_id)
{
    typedef NetGeo<N> G;
    const int ngroups = (c.dv.B + G::G - 1) / G::G;
    dim3 bt(G::NW * 64);
    hipLaunchKernelGGL((k_split<N, 1>), dim3(ngroups, 2), bt, 0, c.stream, c.dv, c.w[net_id], net_id, c.scratch, c.feat);
    hipLaunchKernelGGL((k_split<N, 2>), dim3(ngroups, 4), bt, 0, c.stream, c.dv, c.w[net_id], net_id, c.scratch, c.feat);
    hipLaunchKernelGGL((k_split<N, 3>), dim3(ngroups, 8), bt, 0, c.stream, c.dv, c.w[net_id], net_id, c.scratch, c.feat);
    hipLaunchKernelGGL((k_split<N, 4>), dim3(ngroups), bt, 0, c.stream, c.dv, c.w[net_id], net_id, c.scratch, c.feat);
}

long long split_scratch_floats(int slots)
{
    typedef NetGeo<N> G;
    return (long long)((slots + G::G - 1) / G::G) * SplitGeo<N>::PER_GROUP;
}

void fc(const LaunchCtx &c, int net_id)
{
    unsigned long long *dbgfc = c.dbg ? c.dbg + (size_t)c.dv.B * 16 : nullptr;
    if (c.model == 1) {
        typedef ResGeo<N> G;
        dim3 gf((c.dv.B + 15) / 16, G::NSPLIT), bf(G::FCW * 64);
        hipLaunchKernelGGL(k_fc<G>, gf, bf, 0, c.stream, c.dv, c.w[net_id], net_id, (const float *)c.feat, dbgfc);
    } else {
        typedef NetGeo<N> G;
        dim3 gf((c.dv.B + 15) / 16, G::NSPLIT), bf(G::FCW * 64);
        hipLaunchKernelGGL(k_fc<G>, gf, bf, 0, c.stream, c.dv, c.w[net_id], net_id, (const float *)c.feat, dbgfc);
    }
}

void step(const LaunchCtx &c, int rootN, int do_select)
{
    dim3 g((c.d.B + 3) / 4), b(256);
    const size_t lds = (size_t)(c.d.S + 2) * sizeof(double);   // sqrt table
    if (c.synthetic)
        hipLaunchKernelGGL((k_step<N, true>), g, b, lds, c.stream, c.d, rootN, do_select);
    else
        hipLaunchKernelGGL((k_step<N, false>), g, b, lds, c.stream, c.d, rootN, do_select);
}

void step_vl(const LaunchCtx &c, int sims_done, int nb_next)
{
    dim3 g((c.d.B + 3) / 4), b(256);
    const size_t lds = (size_t)(c.d.S + 2) * sizeof(double);   // sqrt table
    if (c.synthetic)
        hipLaunchKernelGGL((k_step_vl<N, true>), g, b, lds, c.stream, c.d, sims_done, nb_next);
    else
        hipLaunchKernelGGL((k_step_vl<N, false>), g, b, lds, c.stream, c.d, sims_done, nb_next);
}

void root_cache(const LaunchCtx &c)
{
    dim3 g((c.d.B + 3) / 4), b(256);
    hipLaunchKernelGGL(k_root_cache<N>, g, b, 0, c.stream, c.d);
}

void move(const LaunchCtx &c)
{
    dim3 g((c.d.B + 3) / 4), b(256);
    hipLaunchKernelGGL(k_move<N>, g, b, 0, c.stream, c.d);
}

void eval_tail_l(const LaunchCtx &c, int count, float *pol, float *val)
{
    dim3 g((count + 3) / 4), b(256);
    hipLaunchKernelGGL(k_eval_tail<N>, g, b, 0, c.stream, c.dv, count, pol, val);
}
}   // namespace

const SizeOps *AZ_CAT(az_size_ops_, AZ_N)()
{
    static const SizeOps ops = {trunk, trunk_split, split_scratch_floats, fc, step, step_vl, root_cache, move, eval_tail_l};
    return &ops;
}
