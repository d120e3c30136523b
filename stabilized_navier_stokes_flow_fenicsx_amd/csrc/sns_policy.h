// The policy of the AMG hierarchy in ONE place (round 5): every size threshold and every sweep schedule --
// which levels take aggregate blocks, how many sweeps a level runs before and after its coarse-grid correction, where the
// hierarchy ends and how its last level is solved, which level a partitioned run replicates from, which levels run as a
// hipGraph.  Pure host code over (options, a few global counts): no HIP, no handle, so that the table can be unit-tested on the
// CPU (sns_host_cycle_policy, tests/test_host.py) and every rank of a partitioned run answers alike by construction.
// The handle-side predicates of csrc/sns_cycle.hip / sns_setup.hip gather the facts and ask here; the numbers live nowhere else.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>

#include "sns.h"

namespace sns {
namespace policy {

// ---- hierarchy shape -----------------------------------------------------------------------------------------------------------
// rows at or below which a level >= 1 ends the hierarchy (it is solved directly): the dense level of round 4 (amg_dense_rows,
// capped at 4096 rows = a 16 k x 16 k inverse) or the one-workgroup inverse of rounds 1-3 (amg_coarse_size)
inline int coarsest_rows(const sns_options& o) { return std::max(o.amg_coarse_size, std::min(o.amg_dense_rows, 4096)); }
// the one-workgroup inverse with partial pivoting takes at most this many nodes
inline int small_inverse_rows(const sns_options& o) { return std::max(o.amg_coarse_size, 40); }
enum CoarsestKind { COARSEST_SMALL_INVERSE = 0, COARSEST_BLOCKED_INVERSE = 1, COARSEST_SWEEPS = 2 };
inline CoarsestKind coarsest_kind(const sns_options& o, int64_t rows) {
    if (rows <= small_inverse_rows(o)) return COARSEST_SMALL_INVERSE;
    if (rows <= o.amg_dense_rows) return COARSEST_BLOCKED_INVERSE;
    return COARSEST_SWEEPS;                              // amg_max_levels reached with a large last level: 1 + 8 nodal sweeps
}
// a partitioned run replicates the hierarchy from the first level >= 1 with at most amg_replicate_rows GLOBAL rows (and more
// than the small inverse takes; `fits`: the replicated level fits the scratch vectors of every rank's fine level)
inline bool replicate_from(const sns_options& o, int level, int64_t rows_global, bool fits) {
    return level >= 1 && o.amg_replicate_rows > 0 && rows_global <= (int64_t)o.amg_replicate_rows &&
           rows_global > (int64_t)small_inverse_rows(o) && fits;
}
// without a replicated tail the distributed coarsest level is all-gathered into one dense system of at most this many dofs
constexpr int DISTRIBUTED_DENSE_MAX_DOFS = 640;
// first level (>= 1) with at most this many rows: it and everything below are launch-bound and run as ONE hipGraph
// (10 M tets: level 2, 36 k rows; level 1 has 218 k rows = 46 us per sweep)
constexpr int GRAPH_MAX_ROWS = 150000;

// ---- smoother kind ---------------------------------------------------------------------------------------------------------------
// aggregate blocks on the FINE level: always with amg_block_smooth = 2; with 1 on a partitioned handle whose share of the fine
// level is at most amg_block_fine_rows rows per rank -- the latency-bound strong split
inline bool fine_blocks(const sns_options& o, int nranks, int64_t rows_global_fine) {
    if (o.amg_block_smooth >= 2) return true;
    if (o.amg_block_smooth < 1 || o.amg_block_fine_rows <= 0) return false;
    if (nranks < 2 || rows_global_fine <= 0) return false;
    return rows_global_fine <= (int64_t)o.amg_block_fine_rows * nranks;
}
// ... on a level in general: the options allow it and the level is small enough per rank (amg_block_max_rows, 0 = no limit;
// `sharing` = ranks the level's rows are spread over: 1 for a serial or replicated level)
inline bool blocks_allowed(const sns_options& o, int64_t rows_global, int sharing) {
    if (o.amg_block_smooth <= 0 || o.amg_f32_matrix == 0 || o.pc_type != SNS_PC_AMG) return false;
    return !(o.amg_block_max_rows > 0 && rows_global > (int64_t)o.amg_block_max_rows * std::max(1, sharing));
}

// ---- sweep schedule -------------------------------------------------------------------------------------------------------------
// extra sweeps on level 2 and below for LARGE problems (amg_nu_scale_with_size): the plain-aggregation V-cycle loses convergence
// with its depth and on a big mesh those levels cost next to nothing (profiles/r3_deep_sweeps.txt: 81 M tets 73 / 82 -> 53 / 57
// iterations).  depth_equiv: the depth as rounds 1-3 counted it; small_aggregates: the first coarsening keeps more than one row in
// six, i.e. an unstructured mesh (config 4u: 71 -> 58 iterations per Newton step with the first tier).
struct ExtraSweeps { int l2 = 0, deep = 0; };
inline ExtraSweeps extra_sweeps(const sns_options& o, int64_t rows_global_fine, int depth_equiv, bool small_aggregates) {
    ExtraSweeps e;
    if (!o.amg_nu_scale_with_size) return e;
    if (rows_global_fine >= 20000000) { e.l2 = 6; e.deep = 10; }               // 192 M tets: 63 / 71 -> 55 / 66, -10 % time
    else if (rows_global_fine >= 8000000) { e.l2 = 4; e.deep = 6; }
    else if (rows_global_fine >= 2500000 || (depth_equiv >= 7 && small_aggregates)) { e.l2 = 2; e.deep = 2; }
    return e;
}
// depth of the hierarchy as rounds 1-3 counted it: a hierarchy that ends in the dense level of round 4 would have gone on for
// ~log5(rows / amg_coarse_size) more levels
inline int depth_equivalent(const sns_options& o, int nlevels_smoothed_and_last, bool last_blocked_inverse, int64_t last_rows) {
    int n = nlevels_smoothed_and_last;
    if (last_blocked_inverse && last_rows > o.amg_coarse_size)
        n += (int)std::ceil(std::log((double)last_rows / std::max(1, o.amg_coarse_size)) / std::log(5.0));
    return n;
}
// sweeps per half cycle of level `ll` (its number in the hierarchy as coarsened: the replicated copy is not a new level).
// Nodal blocks: the fine level is the expensive one (1 sweep); levels 1 and 2 are cheap and are where plain aggregation needs
// the smoothing (4 and 6); levels >= 3 are launch-bound (2).  Aggregate blocks: one sweep is worth about two nodal sweeps.
inline int level_nu(const sns_options& o, int ll, bool blocks, const ExtraSweeps& e) {
    if (blocks && ll >= 1) {
        if (ll >= 3) return std::max(1, o.amg_bnu_deep) + (e.deep + 1) / 2;
        if (ll == 2) return std::max(1, o.amg_bnu_l2) + (e.l2 + 1) / 2;
        return std::max(1, o.amg_bnu_l2);
    }
    int nu = std::max(1, o.amg_nu);
    if (ll >= 3 && o.amg_nu_deep > 0) nu = o.amg_nu_deep + e.deep;
    else if (ll == 2 && o.amg_nu_l2 > 0) nu = o.amg_nu_l2 + e.l2;
    else if (ll >= 1 && o.amg_nu_coarse > 0) nu = o.amg_nu_coarse;
    return nu;
}
// sweeps before / after the coarse-grid correction (the first pre-sweep is w S b from the zero guess).  Level 1: asymmetric --
// post-smoothing is the more valuable half under a piecewise-constant prolongation: 1 + amg_bnu_l1 with aggregate blocks,
// 1 + (nu + 2) with nodal blocks -- UNLESS its sweeps are rank-local (a partitioned level whose sweeps do not see the neighbours'
// iterate: nu + nu there, 1 + 6 costs 8-11 % more iterations).  A PARTITIONED level 1 with exact global sweeps (exact_partitioned)
// runs one post-sweep more, 1 + (amg_bnu_l1 + 1): its rows per rank are few, a sweep costs ~11 us, and the 4- / 8-way split of the
// 10 M-tet duct needs 91 / 89 instead of 98 / 95 iterations for it (profiles/r5_l1_schedules.txt; 4 + 4 rank-local: 91 / 97).
// amg_nu_l1_pre / _post fix the counts.
struct Sweeps { int pre = 1, post = 1; };
inline Sweeps level_sweeps(const sns_options& o, int ll, bool blocks, bool rank_local_sweeps, int nu, bool exact_partitioned = false) {
    Sweeps s;
    s.pre = s.post = nu;
    if (ll == 1) {
        if (blocks) {
            if (!rank_local_sweeps) { s.pre = 1; s.post = std::max(1, o.amg_bnu_l1) + (exact_partitioned ? 1 : 0); }
        } else if (o.amg_nu_l1_pre == 0 && o.amg_nu_l1_post == 0 && !rank_local_sweeps && nu >= 2) {
            s.pre = 1;
            s.post = nu + 2;
        }
        if (o.amg_nu_l1_pre > 0) s.pre = o.amg_nu_l1_pre;
        if (o.amg_nu_l1_post > 0) s.post = o.amg_nu_l1_post;
    }
    return s;
}

// ---- the table: what a hierarchy of the given shape runs ------------------------------------------------------------------------
// rows[l]: GLOBAL rows of level l as held (a replicated level: its rows); rep_level: first replicated level (0: none; the level
// before it is only the source of the copy and is not cycled); windows: the transport reads ghost entries from receive windows
// (peer / team) and amg_exact_sweeps applies; has_blocks[l]: the level's aggregates have at most 8 members (its smoother blocks
// exist).  kind: SNS_LEVEL_* of include/sns.h.
struct LevelRow { int kind = 0, pre = 0, post = 0, exact = 0, cycled = 1; };
inline void cycle_table(const sns_options& o, int nranks, bool windows, int nlevels, const int64_t* rows, int rep_level,
                        int64_t rows_global_l1, const bool* has_blocks, LevelRow* out) {
    const int last = nlevels - 1;
    const bool part = nranks > 1;
    const int n_as_coarsened = nlevels - (rep_level > 0 ? 1 : 0);
    const CoarsestKind ck = coarsest_kind(o, rows[last]);
    const bool small_agg = rows_global_l1 > 0 && (double)rows[0] < 6.0 * (double)rows_global_l1;
    const ExtraSweeps e = extra_sweeps(o, rows[0], depth_equivalent(o, n_as_coarsened, ck == COARSEST_BLOCKED_INVERSE && nlevels > 1, rows[last]),
                                       small_agg);
    for (int l = 0; l < nlevels; ++l) {
        LevelRow r;
        const bool replicated = rep_level > 0 && l >= rep_level;
        const int ll = replicated ? l - 1 : l;
        if (rep_level > 0 && l == rep_level - 1) { r.cycled = 0; out[l] = r; continue; }
        if (l == last && nlevels > 1) {
            r.kind = ck == COARSEST_BLOCKED_INVERSE ? SNS_LEVEL_DIRECT_BLOCKED : ck == COARSEST_SMALL_INVERSE ? SNS_LEVEL_DIRECT
                                                                                                           : SNS_LEVEL_SWEEPS_ONLY;
            if (part && rep_level == 0 && ck != COARSEST_SWEEPS) r.kind = SNS_LEVEL_DIRECT;      // the all-gathered dense system
            out[l] = r;
            continue;
        }
        const int sharing = (part && !replicated) ? nranks : 1;
        bool blocks = has_blocks[l] && blocks_allowed(o, rows[l], sharing);
        if (l == 0) blocks = blocks && fine_blocks(o, nranks, rows[0]);
        const bool partitioned_level = part && !replicated;
        r.exact = (partitioned_level && l >= 1 && windows && o.halo_windows && o.amg_exact_sweeps && blocks && o.amg_fused_post &&
                   o.amg_fuse_restrict != 0 && !(o.amg_sweep_exchange_rows > 0))
                      ? 1 : 0;
        const int nu = level_nu(o, ll, blocks, e);
        const Sweeps s = level_sweeps(o, ll, blocks, partitioned_level && !r.exact, nu, r.exact != 0);
        r.kind = blocks ? SNS_LEVEL_AGGREGATE_BLOCKS : SNS_LEVEL_NODAL_BLOCKS;
        r.pre = s.pre;
        r.post = s.post;
        out[l] = r;
    }
}

}  // namespace policy
}  // namespace sns
