// az_kernels.hip -- the size-templated kernels instantiated for ONE board size (compile with -DAZ_N=n).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "az_launch.h"

#ifndef AZ_N
#error "compile with -DAZ_N=<board size>"
#endif
#define AZ_CAT2(a, b) a##b
#define AZ_CAT(a, b) AZ_CAT2(a, b)

namespace {
constexpr int N = AZ_N;
const bool g_tile_split = !(getenv("AZ_TILE_SPLIT") && getenv("AZ_TILE_SPLIT")[0] == '0');

void trunk(const LaunchCtx &c, int net_id)
{
    if (c.model == 1) {
        typedef ResGeo<N> G;
        dim3 gt((c.dv.B + G::G - 1) / G::G), bt(G::NW * 64);
#ifndef AZ_EXPERIMENT     // experiment builds (another wave count for the float32 trunk) leave the emulated trunks out
        if (c.emul == EMUL_BF16X3) hipLaunchKernelGGL((k_trunk_res_emul<N, EMUL_BF16X3>), gt, dim3(ResGeoEmul<N>::NW * 64), 0, c.stream, c.dv, c.rw[net_id], net_id, c.feat);
        else if (c.emul == EMUL_F16X2) hipLaunchKernelGGL((k_trunk_res_emul<N, EMUL_F16X2>), gt, dim3(ResGeoEmul<N>::NW * 64), 0, c.stream, c.dv, c.rw[net_id], net_id, c.feat);
        else
#endif
        hipLaunchKernelGGL(k_trunk_res<N>, gt, bt, 0, c.stream, c.dv, c.rw[net_id], net_id, c.feat);
    } else {
        typedef NetGeo<N> G;
        const int ngroups = (c.dv.B + G::G - 1) / G::G;
#ifndef AZ_EXPERIMENT
        if (c.emul == EMUL_BF16X3) {
            hipLaunchKernelGGL((k_trunk_emul<N, EMUL_BF16X3>), dim3(ngroups), dim3(G::NW * 64), 0, c.stream, c.dv, c.w[net_id], net_id, c.feat, c.dbg);
            return;
        }
        if (c.emul == EMUL_F16X2) {
            hipLaunchKernelGGL((k_trunk_emul<N, EMUL_F16X2>), dim3(ngroups), dim3(G::NW * 64), 0, c.stream, c.dv, c.w[net_id], net_id, c.feat, c.dbg);
            return;
        }
#endif
        dim3 gt(AZ_SEQ == 0 ? (ngroups < 256 ? ngroups : 256) : (ngroups + AZ_SEQ - 1) / AZ_SEQ), bt(G::NW * 64);
        hipLaunchKernelGGL(k_trunk<N>, gt, bt, 0, c.stream, c.dv, c.w[net_id], net_id, c.feat, c.dbg);
    }
}

void trunk_split(const LaunchCtx &c, int net_id)
{
    if (c.model == 1) {
        typedef ResGeo<N> G;
        const int ngroups = (c.dv.B + G::G - 1) / G::G;
        dim3 bt(G::NW * 64);
        const ResWeights &w = c.rw[net_id];
        if (g_tile_split) {    // by cell tiles: six launches of 4-wave workgroups (az_net.h k_tile_res)
            const dim3 gt(ngroups, G::MT), b4(256);
            hipLaunchKernelGGL((k_tile_res<N, 0>), gt, b4, 0, c.stream, c.dv, w.stem, w.stemb, w.blk[0], w.blkb[0], net_id, c.scratch, c.feat);
            hipLaunchKernelGGL((k_tile_res<N, 1>), gt, b4, 0, c.stream, c.dv, w.blk[1], w.blkb[1], nullptr, nullptr, net_id, c.scratch, c.feat);
            hipLaunchKernelGGL((k_tile_res<N, 2>), gt, b4, 0, c.stream, c.dv, w.blk[2], w.blkb[2], nullptr, nullptr, net_id, c.scratch, c.feat);
            hipLaunchKernelGGL((k_tile_res<N, 1>), gt, b4, 0, c.stream, c.dv, w.blk[3], w.blkb[3], nullptr, nullptr, net_id, c.scratch, c.feat);
            hipLaunchKernelGGL((k_tile_res<N, 2>), gt, b4, 0, c.stream, c.dv, w.blk[4], w.blkb[4], nullptr, nullptr, net_id, c.scratch, c.feat);
            hipLaunchKernelGGL((k_tile_res<N, 3>), gt, b4, 0, c.stream, c.dv, w.blk[5], w.blkb[5], w.hd, w.hdb, net_id, c.scratch, c.feat);
            return;
        }
        hipLaunchKernelGGL((k_split_res<N, 0>), dim3(ngroups, 4), bt, 0, c.stream, c.dv, w.stem, w.stemb, net_id, c.scratch, c.feat);
        for (int blk = 0; blk < 3; blk++) {
            hipLaunchKernelGGL((k_split_res<N, 1>), dim3(ngroups, 4), bt, 0, c.stream, c.dv, w.blk[2 * blk], w.blkb[2 * blk], net_id, c.scratch, c.feat);
            hipLaunchKernelGGL((k_split_res<N, 2>), dim3(ngroups, 4), bt, 0, c.stream, c.dv, w.blk[2 * blk + 1], w.blkb[2 * blk + 1], net_id, c.scratch, c.feat);
        }
        hipLaunchKernelGGL((k_split_res<N, 3>), dim3(ngroups), bt, 0, c.stream, c.dv, w.hd, w.hdb, net_id, c.scratch, c.feat);
        return;
    }
    typedef NetGeo<N> G;
    const int ngroups = (c.dv.B + G::G - 1) / G::G;
    dim3 bt(G::NW * 64);
    if (g_tile_split) {        // by cell tiles: two launches (az_net.h k_tile); AZ_TILE_SPLIT=0 keeps the channel-tile stages
        hipLaunchKernelGGL((k_tile<N, 1>), dim3(ngroups, G::MT), bt, 0, c.stream, c.dv, c.w[net_id], net_id, c.scratch, c.feat);
        hipLaunchKernelGGL((k_tile<N, 2>), dim3(ngroups, G::MT), bt, 0, c.stream, c.dv, c.w[net_id], net_id, c.scratch, c.feat);
        return;
    }
    hipLaunchKernelGGL((k_split<N, 1>), dim3(ngroups, 2), bt, 0, c.stream, c.dv, c.w[net_id], net_id, c.scratch, c.feat);
    hipLaunchKernelGGL((k_split<N, 2>), dim3(ngroups, 4), bt, 0, c.stream, c.dv, c.w[net_id], net_id, c.scratch, c.feat);
    hipLaunchKernelGGL((k_split<N, 3>), dim3(ngroups, 8), bt, 0, c.stream, c.dv, c.w[net_id], net_id, c.scratch, c.feat);
    hipLaunchKernelGGL((k_split<N, 4>), dim3(ngroups), bt, 0, c.stream, c.dv, c.w[net_id], net_id, c.scratch, c.feat);
}

long long split_scratch_floats(int slots, int model)
{
    if (model == 1) return (long long)((slots + ResGeo<N>::G - 1) / ResGeo<N>::G) * ResSplitGeo<N>::PER_GROUP;
    typedef NetGeo<N> G;
    return (long long)((slots + G::G - 1) / G::G) * SplitGeo<N>::PER_GROUP;
}

template <class G>
void fc_t(const LaunchCtx &c, int net_id, const NetWeights &w, unsigned long long *dbgfc)
{
    // few board rows: one tile per workgroup (the latency shape); many: eight per workgroup in four rounds of two (the
    // throughput shape: 3 workgroups per board row at n = 15).  AZ_FC_SHAPE=1|8 forces one of them.
    const int rows = (c.dv.B + 15) / 16;
    static const int forced = getenv("AZ_FC_SHAPE") ? atoi(getenv("AZ_FC_SHAPE")) : 0;
    if (forced == 1 || (forced < 8 && rows <= 4)) {
        dim3 gf(rows, G::NTP + 4), bf(256);
        hipLaunchKernelGGL((k_fc<G, 1, 1>), gf, bf, 0, c.stream, c.dv, w, net_id, (const float *)c.feat, dbgfc);
    } else if (forced == 16) {      // experiment: 16 tiles per workgroup, 4 rounds of 4 (16 waves)
        dim3 gf(rows, (G::NTP + 15) / 16 + 1), bf(1024);
        hipLaunchKernelGGL((k_fc<G, 4, 4>), gf, bf, 0, c.stream, c.dv, w, net_id, (const float *)c.feat, dbgfc);
    } else if (forced == 17) {      // experiment: 16 tiles per workgroup, 8 rounds of 2 (8 waves)
        dim3 gf(rows, (G::NTP + 15) / 16 + 1), bf(512);
        hipLaunchKernelGGL((k_fc<G, 2, 8>), gf, bf, 0, c.stream, c.dv, w, net_id, (const float *)c.feat, dbgfc);
    } else {
        dim3 gf(rows, (G::NTP + 7) / 8 + 1), bf(512);
        hipLaunchKernelGGL((k_fc<G, 2, 4>), gf, bf, 0, c.stream, c.dv, w, net_id, (const float *)c.feat, dbgfc);
    }
}

void fc(const LaunchCtx &c, int net_id)
{
    unsigned long long *dbgfc = c.dbg ? c.dbg + (size_t)c.dv.B * 16 : nullptr;
    if (c.model == 1) fc_t<ResGeo<N>>(c, net_id, c.w[net_id], dbgfc);
    else fc_t<NetGeo<N>>(c, net_id, c.w[net_id], dbgfc);
}

void step(const LaunchCtx &c, int rootN, int do_select)
{
    dim3 g((c.d.B + AZ_STEP_WAVES - 1) / AZ_STEP_WAVES), b(AZ_STEP_WAVES * 64);
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

constexpr bool HAS_SEARCH = N <= 7;      // the LDS-resident tree needs (S + 1) * n*n * 16 B per game
constexpr bool HAS_SEARCH2 = N <= 5;     // two games per workgroup: beyond 5x5 the rows of two games never fit beside the image

template <int GP, bool SY, bool RES>
int search_prepare_t(int S)
{
    if constexpr (HAS_SEARCH) {
        const size_t dyn = (size_t)GP * (S + 1) * N * N * sizeof(Edge);
        const void *fns[2] = {reinterpret_cast<const void *>(&k_search<N, GP, SY, false, RES>),
                              reinterpret_cast<const void *>(&k_search<N, GP, SY, GP == 2 && !SY, RES>)};     // [1]: the tile-subset variant (search_t)
        for (const void *fn : fns) {
            hipFuncAttributes at;
            if (hipFuncGetAttributes(&at, fn) != hipSuccess) return 0;
            if (at.sharedSizeBytes + dyn > 160u * 1024u) return 0;
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) return 0;
        }
        return 1;
    } else {
        return 0;
    }
}

int search_prepare(int S, int games, int synthetic, int model)
{
    const bool res = model == 1 && !synthetic;         // the synthetic evaluator has no net: the plain kernels serve it
    if (games == 2) {
        if constexpr (HAS_SEARCH2)
            return synthetic ? search_prepare_t<2, true, false>(S) : (res ? search_prepare_t<2, false, true>(S) : search_prepare_t<2, false, false>(S));
        return 0;
    }
    if (games == 1) return synthetic ? search_prepare_t<1, true, false>(S) : (res ? search_prepare_t<1, false, true>(S) : search_prepare_t<1, false, false>(S));
    return 0;
}

template <int GP>
void search_t(const LaunchCtx &c)
{
    if constexpr (HAS_SEARCH) {
        dim3 g((c.d.B + GP - 1) / GP), b(AZ_NW * 64);
        const size_t dyn = (size_t)GP * c.d.R * N * N * sizeof(Edge);
        static const int ts_env = getenv("AZ_SEARCH_TS") ? atoi(getenv("AZ_SEARCH_TS")) : -1;      // experiment: force the variant
        const bool ts = GP == 2 && (ts_env >= 0 ? ts_env != 0 : (c.d.cache || c.d.reuse));      // many iterations with one game waiting for the net: compute its tiles only
        if (c.synthetic)
            hipLaunchKernelGGL((k_search<N, GP, true, false, false>), g, b, dyn, c.stream, c.d, c.w[0], c.w[1], c.dbg, NoWeights{}, NoWeights{});
        else if (c.model == 1) {
            if (ts) hipLaunchKernelGGL((k_search<N, GP, false, GP == 2, true>), g, b, dyn, c.stream, c.d, c.w[0], c.w[1], c.dbg, c.rw[0], c.rw[1]);
            else hipLaunchKernelGGL((k_search<N, GP, false, false, true>), g, b, dyn, c.stream, c.d, c.w[0], c.w[1], c.dbg, c.rw[0], c.rw[1]);
        } else {
            if (ts) hipLaunchKernelGGL((k_search<N, GP, false, GP == 2, false>), g, b, dyn, c.stream, c.d, c.w[0], c.w[1], c.dbg, NoWeights{}, NoWeights{});
            else hipLaunchKernelGGL((k_search<N, GP, false, false, false>), g, b, dyn, c.stream, c.d, c.w[0], c.w[1], c.dbg, NoWeights{}, NoWeights{});
        }
    }
}

void search(const LaunchCtx &c, int games)
{
    if constexpr (HAS_SEARCH2) {
        if (games == 2) { search_t<2>(c); return; }
    }
    search_t<1>(c);
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
    static const SizeOps ops = {trunk, trunk_split, split_scratch_floats, fc, step, step_vl, root_cache, search_prepare, search, move, eval_tail_l};
    return &ops;
}
