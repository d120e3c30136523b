/*
 * sns.h -- C ABI of the MI355X-native stabilised Stokes / Navier-Stokes hot path.
 *
 * The reference (mungerct/Stabilized_Navier_Stokes_Flow_FEniCSx) has no FFI:
 * its seam is the PETSc-SNES callback pair plus two driver functions in
 * NavierStokes/NavierStokesChannelFlow.py.  Each entry point below replaces
 * what DOLFINx/FFCx/PETSc do behind the cited reference lines; a maintainer
 * binds them with the ctypes stub shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types in signatures
 *     (a HIP stream travels as void*).
 *   - "_host" pointers are host memory, "_dev" pointers are device memory of
 *     the GPU the handle was created on.  The caller owns every buffer it
 *     passes; the handle owns mesh, BSR matrix, preconditioner and scratch.
 *   - dof numbering: 4*node + c, c in {ux,uy,uz,p} (node-blocked P1-P1 mixed
 *     space of create_boundary_conditions, :128-129).
 *   - every function returns 0 on success, <0 on a hard error (SNS_E_*);
 *     non-convergence is NOT an error (the reference prints the reason and
 *     carries on, :297-298): it is reported through *reason, whose sign
 *     convention follows PETSc (>0 converged, <0 diverged).
 *   - calls are stream-ordered on the handle's stream and synchronous at
 *     return unless stated otherwise.  With a communicator attached every call
 *     is collective over all ranks (as every PETSc call at :288-293 is).
 */
#ifndef SNS_H
#define SNS_H

#include <stdint.h>

/* libsns.so is built with -fvisibility=hidden: the dynamic symbol table holds exactly the functions declared below */
#if defined(__GNUC__) || defined(__clang__)
#define SNS_API __attribute__((visibility("default")))
#else
#define SNS_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sns_ctx* sns_handle;

/* error codes */
#define SNS_OK            0
#define SNS_E_ARG        -1   /* bad argument / shape mismatch            */
#define SNS_E_HIP        -2   /* HIP runtime error (see sns_last_error)   */
#define SNS_E_STATE      -3   /* call order (e.g. solve before assemble)  */
#define SNS_E_MESH       -4   /* degenerate / inverted input mesh         */
#define SNS_E_COMM       -5   /* RCCL / peer-transport error              */

/* weak forms */
#define SNS_FORM_STOKES   0   /* setup_stokes_weak_form      :160-172 */
#define SNS_FORM_NS       1   /* define_navier_stokes_form   :220-251 */

/* Krylov methods (snes_ksp_type :77, petsc_options :198-202) */
#define SNS_KSP_BICGSTAB  0
#define SNS_KSP_FGMRES    1
#define SNS_KSP_TFQMR     2   /* the reference's snes_ksp_type (:77) */

/* preconditioners */
#define SNS_PC_NONE       0
#define SNS_PC_BJACOBI    1   /* 4x4 nodal block Jacobi                               */
#define SNS_PC_AMG        2   /* aggregation AMG, block-Jacobi smoothing (per-rank)   */

/* converged reasons (PETSc numbering, so logs read like the reference's) */
#define SNS_KSP_CONVERGED_RTOL        2
#define SNS_KSP_CONVERGED_ATOL        3
#define SNS_KSP_DIVERGED_ITS         -3
#define SNS_KSP_DIVERGED_BREAKDOWN   -5
#define SNS_KSP_DIVERGED_NANORINF    -9
/* not a PETSc reason and never the final reason of a solve: what sns_get_counters reports (out[6]) for a FIRST attempt that was
 * ended by the stagnation watch of amg_retry_damping and then retried */
#define SNS_KSP_STALLED              -100
#define SNS_SNES_CONVERGED_FNORM_ABS        2
#define SNS_SNES_CONVERGED_FNORM_RELATIVE   3
#define SNS_SNES_CONVERGED_SNORM_RELATIVE   4
#define SNS_SNES_DIVERGED_LINEAR_SOLVE     -3
#define SNS_SNES_DIVERGED_FNORM_NAN        -4
#define SNS_SNES_DIVERGED_MAX_IT           -5
#define SNS_SNES_DIVERGED_LINE_SEARCH      -6

/* solver knobs; sns_default_options() fills the reference's values.  The shape of the AMG hierarchy (amg_max_levels,
 * amg_coarse_size, amg_agg_size, amg_replicate_rows) is fixed when it is built, lazily at the first sns_pc_setup or
 * solve (after sns_attach_comm, if any): those four must be set before that; every other field can be changed at any
 * time with sns_set_options. */
typedef struct {
    double reynolds;        /* Re, nu = 1/Re                         :223            */
    int    ksp_type;        /* SNS_KSP_*                             :77,:199        */
    int    pc_type;         /* SNS_PC_*                              :200            */
    double ksp_rtol;        /* 1e-8                                  :283            */
    double ksp_atol;        /* 1e-50 (PETSc default)                                 */
    int    ksp_max_it;      /* 10000 (PETSc default)                                 */
    int    gmres_restart;   /* 30 (PETSc default)                                    */
    double snes_rtol;       /* 1e-8                                  :281            */
    double snes_atol;       /* 1e-8                                  :281            */
    double snes_stol;       /* 1e-8 (PETSc default)                                  */
    int    snes_max_it;     /* 30                                    :281            */
    int    amg_max_levels;  /* 12                                                    */
    int    amg_coarse_size; /* stop coarsening at <= this many nodes (dense solve)   */
    int    amg_agg_size;    /* max nodes per aggregate (8)                           */
    int    amg_nu;          /* pre = post smoothing sweeps on the fine level (1)     */
    double amg_omega;       /* block-Jacobi damping (0.8)                            */
    int    monitor;         /* 1: print ||r|| per Krylov/Newton iteration (ksp_monitor / snes_monitor :201,:276) */
    int    corrected_convection; /* 0 = reference as written (dot(u,grad(.)) :241,:247); 1 = (u.grad)(.) */
    int    amg_f32_matrix;  /* storage of the matrix copy the AMG smoother/residual passes read (vectors, D^-1, ALL
                               arithmetic and the Krylov operator stay fp64): 0 = the fp64 operator itself,
                               1 = fp32 copy, 2 (default) = fp16 copy with one fp32 scale per dof row (row-max
                               normalisation; relative perturbation of the preconditioner's matrix <= 2^-11) */
    int    amg_nu_coarse;   /* smoothing sweeps on level 1 (and deeper unless overridden) (4; 0 = same as amg_nu):
                               coarse sweeps are cheap and plain aggregation needs them */
    int    amg_nu_deep;     /* sweeps on levels >= 3 (2; 0 = same as amg_nu_coarse): these levels are launch-bound */
    int    amg_nu_l2;       /* sweeps on level 2 (6; 0 = same as amg_nu_coarse) */
    int    amg_sweep_exchange_rows; /* multi-GPU: AMG levels with at most this many rows per rank exchange the ghost
                               iterate before EVERY smoother sweep (exact global block-Jacobi) instead of smoothing
                               rank-locally; 0 (default) = never.  Fewer iterations, 2*nu instead of 1 exchange per
                               level and cycle */
    int    amg_replicate_rows; /* multi-GPU: the first AMG level (>= 1) with at most this many GLOBAL rows, and all below,
                               are held and cycled redundantly by every rank (values all-gathered at setup, one
                               all-gather of the right-hand side per cycle, no exchanges below); 0 = off.  Default 65536 */
    int    amg_post_exchange; /* multi-GPU: 1 (default) = on levels with ONE post-smoothing sweep (the fine level) one more
                               ghost exchange after the coarse-grid correction, which makes that sweep the exact
                               global block-Jacobi sweep */
    int    assembly_fused;  /* 1: scratch-free Jacobian assembly (each BSR block recomputed by its owner lane) when the
                               state satisfies the Dirichlet data; 0: always the staged element kernel + gather */
    double stokes_viscosity; /* 2-D handles only: viscosity of the Stokes form (1.0: DFG_2D_Validation.py:107-110;
                               nu = 1/Re: LidDrivenNavierStokesFlow.py:96-104) */
    double stokes_beta;     /* 2-D handles only: mu_T = stokes_beta * h^2 (0.2: DFG_2D_Validation.py:104-106;
                               a0/(4 nu), a0 = 1/3: LidDrivenNavierStokesFlow.py:98-100) */
    int    amg_nu_l1_pre;   /* sweeps BEFORE the coarse-grid correction on level 1; the first one is omega D^-1 b from the zero
                               guess, no matrix pass.  0 (default) = automatic, see amg_nu_l1_post */
    int    amg_nu_l1_post;  /* sweeps AFTER the coarse-grid correction on level 1.  Both 0 (default) = automatic: a single-GPU
                               handle runs 1 + (amg_nu_coarse + 2) = 1 + 6 -- post-smoothing is the more valuable half under a
                               piecewise-constant prolongation: the iterations of 4 + 4 with one level-1 matrix pass less
                               (-3 % per Newton iteration on the 10 M-tet duct, neutral on configs 3 / 4, Stokes +1 iteration);
                               a partitioned handle runs amg_nu_coarse + amg_nu_coarse (its post-sweeps are rank-local and
                               1 + 6 costs 8-11 % more iterations there).  Set both (e.g. 4 and 4) to fix the counts */
    int    amg_retry_damping; /* 1 (default): a Krylov solve that ends in BREAKDOWN or NANORINF under SNS_PC_AMG is retried
                               once from the same guess with every level's block-Jacobi damping scaled by 0.7 (see
                               sns_krylov_solve); 0: the failed reason is reported and that is it, as PETSc does */
    int    amg_retry_stall_its; /* with amg_retry_damping: the first BiCGStab attempt is also ended (SNS_KSP_STALLED) and retried when
                               it STAGNATES: no new best residual norm at all for this many iterations, or a best residual
                               still >= the initial one after this many (100; 0 = breakdown / NaN only).  A solve that converges
                               slowly keeps setting new bests and is never touched.  An over-relaxed smoother makes BiCGStab
                               stagnate far more often than break down outright */
    int    halo_overlap;    /* multi-GPU: 1 (default) = level-0 passes compute their interior rows on a second stream while the
                               halo exchange is in flight and the boundary rows after it; 0 = exchange, then one full pass
                               (same arithmetic per row, bitwise the same result).  The environment variable
                               SNS_NO_OVERLAP=1 forces 0 */
    int    amg_fused_post;  /* 1 (default): the coarse-grid correction and the first post-smoothing sweep of a level are ONE
                               pass over M = A P (fine rows x coarse columns, 0.37x the blocks of A on a tet mesh's fine
                               level; same low-precision format as the level matrix): z = (x1 + P xc) + w Dinv (r1 - M xc).
                               The same linear operator as prolongation + sweep up to rounding.  Needs amg_f32_matrix != 0;
                               0 = prolongation kernel + a full sweep over A */
    int    amg_nu_scale_with_size; /* 1 (default): more sweeps on level 2 and the deeper levels of LARGE problems, where they cost
                               next to nothing and the plain-aggregation V-cycle loses convergence with its depth: from 2.5 M fine
                               rows (all ranks together) -- or on a hierarchy of >= 7 levels whose first coarsening keeps more than
                               one row in six, i.e. an unstructured mesh -- amg_nu_l2 + 2 and amg_nu_deep + 2, from 8 M rows + 4 and + 6, from 20 M rows + 6
                               and + 10 (81 M tets on one GPU: 73 / 82 -> 53 / 57 BiCGStab iterations per Newton step, -25 % time;
                               192 M tets: 95 / 108 -> 55 / 66).  0 = the counts as given */
    int    amg_dense_rows;  /* (round 4) the first AMG level >= 1 with at most this many block rows (512 = 2048 dofs by default; a
                               partitioned handle: the first such level of its replicated tail) ends the hierarchy and is solved
                               EXACTLY by one matvec with its explicit inverse (fp32 storage, fp64 accumulation), rebuilt at every
                               numeric setup by a blocked Gauss-Jordan elimination on the fp64 matrix cores (csrc/sns_dense.hip,
                               2 N^3 flops, < 1 ms at N = 2048): one launch instead of the ~15 dependent latency-bound launches
                               of the deepest levels of the V-cycle.  Fixed when the hierarchy is built, like amg_coarse_size.
                               0 = coarsen down to amg_coarse_size nodes as in rounds 1-3 */
    int    amg_block_smooth; /* (round 4) 1 (default): the smoothed levels >= 1 use AGGREGATE-block Jacobi -- the dense block of the
                               <= 8 nodes (32 dofs) that form one node of the next level, inverted at every numeric setup, in place
                               of the 4 x 4 nodal block: x <- x + w B^-1 (b - A x), one launch per sweep like before and worth about
                               two point-block sweeps, so the levels run the shorter schedule below (csrc/sns_block.hip).  The inverse
                               blocks are held in the format of the level's matrix copy (fp16 + one fp32 scale per row by default,
                               2 KiB + 128 B per aggregate; fp32 with amg_f32_matrix = 1).  2: the fine level as well (1 takes it on partitioned handles only, see
                               amg_block_fine_rows).  0: nodal blocks everywhere (rounds 1-3).
                               Needs amg_f32_matrix != 0 and amg_agg_size <= 8; levels where it does not apply keep the nodal
                               blocks and their sweep counts.  Fixed when the hierarchy is built */
    int    amg_bnu_l1;      /* sweeps after the coarse-grid correction on level 1 under amg_block_smooth (3; one sweep before it);
                               a partitioned handle runs amg_bnu_l2 + amg_bnu_l2 there (rank-local post-sweeps, as with amg_nu_l1_*),
                               or, with exact global sweeps (amg_exact_sweeps over a window transport), 1 + (amg_bnu_l1 + 1) */
    int    amg_bnu_l2;      /* sweeps per half cycle on level 2 under amg_block_smooth (4) */
    int    amg_bnu_deep;    /* ... and on levels >= 3 (2: the nodal blocks' count, with the stronger smoother).  amg_nu_scale_with_size adds half of its extra sweeps (rounded up) to both */
    int    amg_ritz_limit;  /* 1 (default): on every level that runs 3 or more sweeps per cycle the damping is also capped by the
                               STABILITY limit of the dominant Ritz values of S A (S = the smoother's block inverse) from 8 Arnoldi
                               steps, w <= min 2 Re(theta) / |theta|^2: the power iteration of rounds 1-3 sees |lambda|max only,
                               and on a convection-dominated coarse level the dominant eigenvalues are complex -- a damping above
                               the limit is amplified by every one of the level's sweeps (jittered 120 x 30 x 30 duct, Re 200:
                               level 1 at w = 0.68 against a limit of 0.45 stalls BiCGStab; oracle/experiments/r4_damping.py).
                               0: the |lambda|max rule alone */
    int    amg_block_max_rows; /* aggregate blocks only on levels with at most this many rows per rank; 0 (default) = no limit.  (While the
                               inverse blocks were fp32 -- 4 KiB per aggregate -- a block sweep cost 1.5x a nodal sweep on a large
                               level and the limit was 8192; in the format of the level's fp16 matrix copy, 2 KiB + row scales, it
                               costs 1.2x and every coarse level gains: 10 M-tet duct 137.5 -> 127-130 ms per Newton iteration) */
    int    amg_block_fine_rows; /* under amg_block_smooth = 1: aggregate blocks on the FINE level too on a PARTITIONED handle (>= 2 ranks)
                               whose fine level has at most this many rows per rank (600000; 0 = never).  The strong split is
                               latency-bound: 15-21 % fewer BiCGStab iterations -- and collectives -- for 6 % more time per
                               iteration (10 M-tet duct on 2 / 4 / 8 ranks: 41 / 43 / 47 -> 35 / 33 / 36 iterations; its 1/8 share on
                               one GPU 19.3-20.2 -> 17.9-18.0 ms per Newton iteration), where a whole mesh on one GPU loses
                               (10 M tets: 42 -> 37 iterations, 129 -> 132 ms; 893 k nodes +2.5 %), which is why a single-GPU
                               handle takes them with amg_block_smooth = 2 only.  Fixed when the hierarchy is built */
    int    amg_fuse_restrict; /* 1 (default): below the fine level the residual r = b - A x, its restriction and the next level's
                               first sweep are ONE launch (k_resid_restrict, csrc/sns_block.hip) instead of two -- every launch down
                               there is 5-8 us of latency.  Same sums in the same order; 0 = the separate kernels; 2 = a single-GPU FINE level as
                               well (measured, no: 0.326 ms against 0.234 + 0.030 for the tuned fine-level residual + restriction) */
    int    halo_windows;    /* (round 5) multi-GPU over the window transports (peer windows; the in-process team): 1 (default) = a halo
                               exchange is ONE launch -- the put into the neighbours' receive windows -- and the level pass that consumes
                               it reads the ghost entries straight from its own window, the waves that meet a ghost column waiting for
                               the neighbours' arrival flags themselves: no unpack kernel, no interior / boundary split, no second
                               stream, no staging copies.  0 = put + wait / unpack into the vector's ghost tail, then the passes of
                               round 4 (the RCCL transport always works that way, with pack + send / recv + unpack) */
    /* (retired in round 5, VERDICT r4 item 6: amg_fine_cycle -- the experimental V(0,1) / V(1,0) fine-level cycles, never the default
       and untested since round 3 --, amg_growth_check -- its value 1 is the rule now, csrc/sns_setup.hip) */
    int    amg_exact_sweeps; /* (round 5) 1 (default): on the window transports the aggregate-block-smoothed PARTITIONED levels >= 1 run
                               the single-GPU schedule (level 1: 1 + (amg_bnu_l1 + 1) sweeps, the coarse-grid correction inside the
                               first post-sweep) with EXACT global sweeps -- a flag wait per sweep, the put rides in the kernel that
                               produces the iterate -- instead of amg_bnu_l2 + amg_bnu_l2 rank-local sweeps with an unfused
                               correction (8-way split of the 10 M-tet duct: 31 / 34 instead of 36 / 36 BiCGStab iterations with 6
                               instead of 9 level-1 launches per cycle).  0, and always over RCCL (an exchange per sweep costs a send / recv group there): round 4's cycle */
} sns_options;

SNS_API void sns_default_options(sns_options* opt);
SNS_API const char* sns_last_error(void);
SNS_API const char* sns_version(void);
/* ABI guard (round 4): the library's SNS_ABI_VERSION and sizeof(sns_options).  sns_options grows at its END between rounds
 * and the fixed-size out-arrays of the getters below have grown (sns_get_counters / sns_get_kernel_times: 4 -> 8 entries in
 * round 3), so a binding built against an older header would pass short buffers: bindings compare both numbers with the
 * header they were written against before making any other call (the ctypes mirror does, _lib.py) and refuse on a mismatch. */
#define SNS_ABI_VERSION 7
SNS_API int sns_abi_version(void);
SNS_API int64_t sns_options_size(void);

/* ---- setup: replaces gmshio.model_to_mesh + functionspace + create_matrix +
 *      locate_dofs_topological/dirichletbc (:111,:127-147,:271-272) ----------
 * points_host  [n_nodes*3]   vertex coordinates
 * tets_host    [n_tets*4]    connectivity, cell-local vertex order PRESERVED
 * bc_mask_host [4*n_nodes]   1 where the dof is Dirichlet-constrained
 * bc_val_host  [4*n_nodes]   prescribed value g (ignored where mask==0)
 * device       HIP device ordinal                                              */
SNS_API int sns_create(sns_handle* out, int32_t n_nodes, int64_t n_tets,
               const double* points_host, const int32_t* tets_host,
               const uint8_t* bc_mask_host, const double* bc_val_host,
               int device, const sns_options* opt);
/* 2-D variant (triangles, LidDrivenFlow/LidDrivenNavierStokesFlow.py:29-30 create_rectangle, Validation_Flow/
 * DFG_2D_Validation.py:28 read_from_msh(gdim=2)): points_host [n_nodes*2], tris_host [n_tris*3].  The handle keeps
 * the node-blocked layout of 4 dofs per node [ux, uy, uz, p]; uz is constrained to 0 by the library (identity rows),
 * bc_mask / bc_val still have 4*n_nodes entries.  On such a handle SNS_FORM_STOKES is the pressure-stabilised Stokes
 * form with (stokes_viscosity, stokes_beta) and SNS_FORM_NS the P1-P1 form with the h-based Tezduyar UGN
 * tau_SUPG / tau_LSIC (LidDrivenNavierStokesFlow.py:123-143 == DFG_2D_Validation.py:141-163) and its exact Gateaux
 * derivative; every other entry point works unchanged.  Single GPU only (no sns_attach_comm).                    */
SNS_API int sns_create_2d(sns_handle* out, int32_t n_nodes, int64_t n_tris,
                  const double* points_host, const int32_t* tris_host,
                  const uint8_t* bc_mask_host, const double* bc_val_host,
                  int device, const sns_options* opt);
SNS_API int sns_destroy(sns_handle h);
SNS_API int sns_set_stream(sns_handle h, void* hip_stream);
SNS_API int sns_set_options(sns_handle h, const sns_options* opt);
SNS_API int sns_get_options(sns_handle h, sns_options* opt);
/* DIAGNOSTIC (round 5): perturb the 3-D NS form of define_navier_stokes_form (:220-251) -- the study of what the reference-held
 * constants (DFG_2D_Validation.py:202-203) can tell apart, tests/test_gpu_2d.py.  c_inverse: C_I of the G-metric tau (:237; 36;
 * 0 drops the 36 nu^2 G:G term), lsic_scale: factor on nu_L (:249; 1; 0 = no LSIC term), pspg_sign: sign of (tau res_M, grad q)
 * (:247; +1), one_point_quadrature != 0: the 1-point centroid rule instead of the 4-point rule of dx(degree 2) (:222).  The values
 * in brackets restore the reference's form.  A perturbed form is assembled by the staged element kernel (Jacobian and residual,
 * exact Gateaux derivative of the perturbed form); the matrix must be re-assembled afterwards.                                 */
SNS_API int sns_set_form_variant(sns_handle h, double c_inverse, double lsic_scale, double pspg_sign, int one_point_quadrature);

/* sizes: n_owned = rows this rank owns, n_local = owned + ghost nodes */
SNS_API int sns_get_sizes(sns_handle h, int32_t* n_local_nodes, int32_t* n_owned_nodes,
                  int64_t* n_tets, int64_t* nnz_blocks);

/* ---- distributed setup (one process per GPU; RCCL over xGMI) ---------------
 * Local mesh = owned nodes first [0,n_owned), ghost nodes after; tets = all
 * tets touching an owned node (redundant ghost-tet assembly: no assembly
 * communication, replaces F.ghostUpdate(ADD,REVERSE) :66 and MatAssembly :75).
 * Neighbour k: we send the owned-node values listed in send_idx[send_ptr[k]..)
 * and receive into ghost slots recv_idx[recv_ptr[k]..) (local node ids).
 * nccl_unique_id: 128 bytes from sns_comm_unique_id on rank 0, broadcast by
 * the caller (torch.distributed).  NULL = no communicator: the handle only
 * learns its owned/ghost split and the caller moves ghost values itself
 * (single-GPU tests of the partitioned path).                                  */
SNS_API int sns_comm_unique_id(char id_out[128]);
SNS_API int sns_attach_comm(sns_handle h, int rank, int nranks, const char nccl_unique_id[128],
                    int32_t n_owned_nodes, int n_neighbors, const int32_t* neighbor_ranks,
                    const int32_t* send_ptr, const int32_t* send_idx,
                    const int32_t* recv_ptr, const int32_t* recv_idx);

/* In-process "team" communicator (TEST TRANSPORT): N ranks = N host threads of one
 * process sharing one GPU; halo exchange / all-reduce / all-gather are emulated with
 * barriers and device copies so the N-rank algorithm can be verified on a 1-GPU box.
 * Every rank (thread) attaches its own handle; all collective calls must then be made
 * concurrently from the N threads.  Never used by bench.py or the drivers.          */
SNS_API int sns_team_create(int nranks, void** team_out);
SNS_API int sns_team_destroy(void* team);
SNS_API int sns_attach_team(sns_handle h, void* team, int rank, int nranks,
                    int32_t n_owned_nodes, int n_neighbors, const int32_t* neighbor_ranks,
                    const int32_t* send_ptr, const int32_t* send_idx,
                    const int32_t* recv_ptr, const int32_t* recv_idx);

/* Peer-window communicator (PRODUCT TRANSPORT for the ranks of ONE node, round 4): no library in the data path.  Every rank
 * owns a window of fine-grained device memory that the other ranks map through HIP IPC; halo exchange, all-reduce and
 * all-gather are small kernels that store into the peers' windows over xGMI and raise sequence flags there (csrc/sns_comm.h).
 * It replaces the same PETSc/MPI traffic as sns_attach_comm (VecGhostUpdate inside MatMult, the KSP's MPI_Allreduce) where the
 * RCCL launch floor (~50 us per send/recv group, ~10 us per all-reduce on this image) dominates: the strong split.
 *   sns_peer_create  : allocate this rank's window (window_bytes, 0 = 64 MiB) on `device`; ipc_handle_out = 64 bytes for the peers
 *   sns_peer_connect : ipc_handles = nranks x 64 bytes in rank order (all-gathered by the caller, e.g. torch.distributed);
 *                      call on every rank, then synchronise the ranks once before the first attach
 *   sns_attach_peer  : as sns_attach_comm (same plan arguments; rank / nranks are the communicator's); collective
 *   sns_peer_disconnect : first half of the teardown, after every handle attached to the communicator is destroyed AND the ranks
 *                      have synchronised: unmaps the other ranks' windows.  Synchronise the ranks once more, then
 *   sns_peer_destroy : frees this rank's window (no peer has it mapped any more: the HIP IPC contract leaves freeing a
 *                      still-mapped allocation undefined).  Without a preceding sns_peer_disconnect it does both halves at once
 * At most 16 ranks.  Every device-side wait is bounded (SNS_PEER_TIMEOUT_MS, default 20000): a late or dead peer turns into
 * SNS_E_COMM at the next host synchronisation instead of a hang.                                                          */
SNS_API int sns_peer_create(int device, int rank, int nranks, int64_t window_bytes, void** peer_out, char ipc_handle_out[64]);
SNS_API int sns_peer_connect(void* peer, const char* ipc_handles);
SNS_API int sns_peer_disconnect(void* peer);
SNS_API int sns_peer_destroy(void* peer);
/* Link check between the REAL ranks of a connected communicator (collective, after sns_peer_connect and one synchronisation of the
 * ranks): `rounds` all-reduces whose contributions depend on rank and round, `rounds` all-gathers of 4096 patterned doubles per rank
 * and `rounds` halo exchanges over a ring between the ranks (2048 nodes per direction; read alternately through the wait / unpack
 * kernel and through the level passes' own path: the wait inside the consumer, the ghost entries straight from the window), every
 * value verified.  The first thing to run on a new machine: a visibility problem of the windows shows up here as SNS_E_COMM with a
 * count instead of as a solve that quietly diverges.                                                                              */
SNS_API int sns_peer_check_links(void* peer, int rounds);
/* Self-test and latency probe of the protocol inside ONE process: nranks (2 or 3) threads with a window, a stream and a
 * communicator end each, wired directly (no IPC), a ring of halo links of `halo_nodes` nodes per direction.  Per collective -- halo
 * exchange, all-reduce (4 doubles), all-gather (2048 doubles per rank) -- `reps` rounds with every payload verified, then `reps`
 * timed rounds of the collective alone: us_out = microseconds per round (max over ranks); then three verified rounds of the long forms
 * (an all-reduce of 40 doubles = two launches, an all-gather of more than three staging chunks).  SNS_E_COMM on a wrong value or a timeout. */
SNS_API int sns_peer_selftest(int device, int nranks, int halo_nodes, int reps, double us_out[3]);
SNS_API int sns_attach_peer(sns_handle h, void* peer, int32_t n_owned_nodes, int n_neighbors, const int32_t* neighbor_ranks,
                    const int32_t* send_ptr, const int32_t* send_idx,
                    const int32_t* recv_ptr, const int32_t* recv_idx);

/* ---- hot path --------------------------------------------------------------*/
/* NonlinearPDE_SNESProblem.F (:51-67): F(w) incl. lifting and F_B = w_B - g.
 * form = SNS_FORM_NS, or SNS_FORM_STOKES for the linear residual A w - b.     */
SNS_API int sns_residual(sns_handle h, int form, const double* w_dev, double* F_dev);
/* NonlinearPDE_SNESProblem.J (:69-75): assemble the Jacobian (BC rows+cols
 * zeroed, unit diagonal) into the handle's BSR matrix; if F_dev != NULL the
 * residual is produced by the same fused element pass.                        */
SNS_API int sns_jacobian(sns_handle h, int form, const double* w_dev, double* F_dev);
/* MatMult with the assembled operator: y = A x (halo exchange inside).        */
SNS_API int sns_spmv(sns_handle h, const double* x_dev, double* y_dev);
/* PCSetUp / PCApply for the current matrix.                                   */
SNS_API int sns_pc_setup(sns_handle h);
SNS_API int sns_pc_apply(sns_handle h, const double* r_dev, double* z_dev);
/* KSPSolve: A x = b with the handle's ksp/pc options; x_dev holds the initial
 * guess on entry.  rnorm = 2-norm of the TRUE residual b - A x at exit.  With
 * amg_retry_damping (default on) a solve that ends in DIVERGED_BREAKDOWN or
 * DIVERGED_NANORINF (or, BiCGStab, stagnates for amg_retry_stall_its iterations: first-attempt
 * reason SNS_KSP_STALLED) under SNS_PC_AMG is retried ONCE from the same guess with every
 * level's block-Jacobi damping scaled by 0.7; the factor stays until the next
 * sns_set_options.  *its then counts both attempts (<= 2 ksp_max_it), *reason is the
 * last attempt's, the first attempt's is out[6] of sns_get_counters.  Running out of
 * iterations (DIVERGED_ITS) is never retried.                                    */
SNS_API int sns_krylov_solve(sns_handle h, const double* b_dev, double* x_dev,
                     int* its, int* reason, double* rnorm);
/* solve_stokes_problem (:197-218): assemble + lift + KSP; U_dev receives U.   */
SNS_API int sns_stokes_solve(sns_handle h, double* U_dev, int* ksp_its, int* reason, double* rnorm);
/* solve_navier_stokes (:268-312): SNES newtonls + bt; w_dev updated in place
 * (snes.solve(None, w) :293).  fnorm_hist (nullable) gets ||F|| per iteration
 * (snes_monitor :276), at most hist_cap entries.                              */
SNS_API int sns_newton_solve(sns_handle h, double* w_dev, int* its, int* reason,
                     int* total_ksp_its, double* fnorm_hist, int hist_cap);

/* ---- introspection (tests, profiling) --------------------------------------*/
/* device pointers of the assembled BSR4 operator (block row-major 4x4)        */
SNS_API int sns_get_bsr(sns_handle h, int32_t* n_rows, int64_t* nnzb, const int32_t** rowptr_dev,
                const int32_t** colind_dev, const double** vals_dev);
/* element-level output of the last sns_jacobian call: Ke [n_tets][a][b][c][d]
 * (16 blocks of 4x4) as produced by the element kernel before the gather.     */
SNS_API int sns_get_element_scratch(sns_handle h, const double** Ke_dev, const double** Fe_dev);
/* copy an internal device array into a caller-owned device buffer of nbytes
 * (exact size required): tests read the assembled operator through this.      */
#define SNS_EXPORT_ROWPTR 0   /* int32 [n_local+1]        */
#define SNS_EXPORT_COLIND 1   /* int32 [nnzb]             */
#define SNS_EXPORT_VALS   2   /* double [nnzb*16]         */
#define SNS_EXPORT_KE     3   /* double [n_tets*256]      */
#define SNS_EXPORT_FE     4   /* double [n_tets*16]       */
SNS_API int sns_export(sns_handle h, int what, void* dst_dev, int64_t nbytes);
/* timing of the phases of the last solve, milliseconds (HIP events)           */
typedef struct {
    double assemble_ms, pc_setup_ms, krylov_ms, spmv_ms_avg;
    int64_t spmv_calls; int ksp_its; int amg_levels;
} sns_timings;
SNS_API int sns_get_timings(sns_handle h, sns_timings* t);
/* debug counters of the LAST Krylov solve: out[0] = host<->device synchronisations (stream / event waits),
 * out[1] = all-reduces, out[2] = neighbour (halo) exchanges; out[3] = Krylov iterations since reset_timings,
 * out[4] = damping retries since reset_timings, out[5] = current damping factor x 1e6 (1000000 = no retry so far),
 * out[6] = reason of the first attempt of the last solve if it was retried (else 0), out[7] = blocks of the fine level's
 * M = A P (fused post-smoothing sweep; 0 until the hierarchy exists) */
SNS_API int sns_get_counters(sns_handle h, int64_t out[8]);
/* communicator of the handle: out[0] = transport (0 none, 1 RCCL, 2 in-process team, 3 peer windows), out[1] = this rank,
 * out[2] = ranks the handle was attached with, out[3] = ranks RCCL itself reports (ncclCommCount; 0 without RCCL):
 * bench.py prints it so that "did RCCL see N ranks" can be read off the result line */
SNS_API int sns_comm_info(sns_handle h, int32_t out[4]);
/* the AMG hierarchy as built (what PETSc's -ksp_view prints of a PCGAMG/PCMG): *nlevels levels (at most 16 reported; a
 * partitioned handle counts its replicated tail copy), per level the rows this rank solves for, the 4x4 blocks of its
 * operator, the sweeps per half cycle level_nu gives it (after amg_nu_scale_with_size) and the block-Jacobi damping in use */
SNS_API int sns_get_hierarchy(sns_handle h, int32_t* nlevels, int64_t rows[16], int64_t blocks[16], int32_t sweeps[16], double omega[16]);
/* the dense coarsest-level solver on its own (tests): Ainv_dev <- inverse of the N x N row-major fp64 matrix A_dev by the blocked
 * Gauss-Jordan elimination of csrc/sns_dense.hip (64 x 64 blocks, v_mfma_f64_16x16x4_f64 rank-64 updates, NO pivoting: meant for
 * matrices whose symmetric part is positive definite, like the level operators).  SNS_E_STATE on a zero / non-finite pivot. */
SNS_API int sns_dense_inverse(int device, int32_t N, const double* A_dev, double* Ainv_dev);
/* the V-cycle as run: per level the smoother / solver kind and the sweeps before and after the coarse-grid correction (the
 * first pre-sweep starts from the zero guess).  Tests restate the cycle from this (oracle/amg_cycle.py).                    */
#define SNS_LEVEL_NODAL_BLOCKS      0   /* damped Jacobi with the 4 x 4 nodal blocks                                         */
#define SNS_LEVEL_AGGREGATE_BLOCKS  1   /* damped Jacobi with the aggregates' dense blocks (amg_block_smooth)                */
#define SNS_LEVEL_DIRECT            2   /* coarsest level: dense inverse by the one-workgroup Gauss-Jordan with pivoting     */
#define SNS_LEVEL_DIRECT_BLOCKED    3   /* coarsest level: dense inverse by the blocked Gauss-Jordan of csrc/sns_dense.hip   */
#define SNS_LEVEL_SWEEPS_ONLY       4   /* coarsest level too large for a direct solve: 1 + 8 nodal-block sweeps              */
SNS_API int sns_get_cycle(sns_handle h, int32_t* nlevels, int32_t kind[16], int32_t nu_pre[16], int32_t nu_post[16]);
SNS_API int sns_reset_timings(sns_handle h);
/* per-launch HIP-event timing of the level-0 k_spmv family inside solves
 * (index = mode: 0 y=Ax, 1 r=b-Ax, 2 Jacobi sweep, 3 y=Ax with fused dot,
 *  4 fused correction + post-sweep on M = A P; 5..7 reserved)                  */
SNS_API int sns_time_kernels(sns_handle h, int on);
SNS_API int sns_get_kernel_times(sns_handle h, double ms_total[8], int64_t calls[8]);
/* raw kernel launchers for micro-benchmarks (bench.py roofline leg): run the
 * kernel `reps` times between two HIP events on the handle's stream and return
 * the average duration in ms.                                                  */
SNS_API int sns_bench_spmv(sns_handle h, const double* x_dev, double* y_dev, int reps, double* ms_avg);
SNS_API int sns_bench_assemble(sns_handle h, int form, const double* w_dev, double* F_dev, int reps, double* ms_avg);
/* one collective of the attached communicator, `reps` times back to back (COLLECTIVE: every rank calls it with the same arguments):
 * which 0 = halo exchange of the assembled operator's plan, 1 = all-reduce of `count` (1..32) doubles, 2 = all-gather of `count`
 * doubles per rank.  Any transport.                                                                                          */
SNS_API int sns_bench_collective(sns_handle h, int which, int count, int reps, double* ms_avg);
/* ---- batched particle tracing (next row after the solve path; replaces the per-seed
 *      solve_ivp(RK45) of NavierStokes/streamtrace.py:208-232, :357-383) ---------------
 * One lane per seed: scipy's RK45 (same tableau, controller and initial step; rtol/atol as
 * given, the reference uses solve_ivp's defaults 1e-3 / 1e-6 and max_step 0.125, t in [0,20]).
 * Velocity = P1 interpolation of vel_dev (n_nodes x 3) in the containing tet, zero outside
 * (velfunc :144-158); reverse != 0 negates it (:160-173).  Terminal events: speed falling
 * below speed_min (1e-6, :175-178) -> status 1; x crossing x_stop upward (forward, 3.7 :180-183)
 * or downward (reverse, 0.13 :185-188) -> status 2; t_end reached -> 0; step underflow -> 3.
 * nbr_dev[4t+a] = tet across the face opposite local vertex a of tet t, -1 on the boundary;
 * seed_tet_dev = a tet containing (or near) each seed.  All pointers are device memory.   */
SNS_API int sns_streamtrace(int32_t n_nodes, int64_t n_tets, const double* pts_dev, const int32_t* tets_dev,
                    const int32_t* nbr_dev, const double* vel_dev, int32_t n_seeds,
                    const double* seeds_dev, const int32_t* seed_tet_dev, int reverse, double t_end,
                    double max_step, double rtol, double atol, double x_stop, double speed_min,
                    double* pos_out_dev, double* t_out_dev, int32_t* status_out_dev,
                    int32_t* steps_out_dev, void* hip_stream);

/* ---- host-only symbolic utilities (no GPU needed; used by sns_create and by
 *      the CPU test-suite) ------------------------------------------------------
 * BSR sparsity pattern of the P1-P1 operator (what create_matrix derives from
 * the dofmap, :272) and the slot -> element-block gather lists of the
 * atomic-free assembly.  Call with NULL outputs to query sizes.               */
SNS_API int sns_host_pattern(int32_t n_nodes, int64_t n_tets, const int32_t* tets_host,
                     int64_t* nnzb_out,
                     int32_t* rowptr_out /*n_nodes+1*/, int32_t* colind_out /*nnzb*/,
                     int64_t* c_ptr_out /*nnzb+1*/, int32_t* c_idx_out /*16*n_tets*/);
/* size-limited greedy aggregation of the first n_active nodes of a pattern;
 * agg_out[n_nodes] gets the aggregate id (-1 for inactive nodes).             */
SNS_API int sns_host_aggregate(int32_t n_nodes, const int32_t* rowptr, const int32_t* colind,
                       int32_t n_active, int max_agg, int32_t* agg_out, int32_t* n_agg_out);
/* (round 5) ... with the nodes' coordinates pts[3 * n_nodes], as the hierarchy build runs it on a 3-D level: the greedy sweep
 * (on an anisotropic cloud along the short edges only), and -- unless its aggregates are as compact as cubes already -- a pairwise
 * aggregation (nodes -> pairs -> quadruples -> octets by closest centroids), which does not depend on how the node numbers run
 * through the mesh; the more compact of the two is returned.  *which_out: 0 = greedy sweep, 1 = pairwise.  pts = NULL: the
 * greedy sweep, = sns_host_aggregate.                                          */
SNS_API int sns_host_aggregate_pts(int32_t n_nodes, const int32_t* rowptr, const int32_t* colind, int32_t n_active, int max_agg,
                           const double* pts, int32_t* agg_out, int32_t* n_agg_out, int32_t* which_out);

/* owned rows (i < n_owned) of a local pattern that reference a ghost column (>= n_owned): the boundary rows of the
 * interior / boundary split of the multi-GPU SpMV -- interior rows are computed while the halo is in flight
 * (MatMult's VecScatter overlap in the reference's PETSc).  rows_out [n_owned], *n_out entries are valid.        */
SNS_API int sns_host_boundary_rows(int32_t n_owned, const int32_t* rowptr, const int32_t* colind,
                           int32_t* rows_out, int32_t* n_out);

/* The policy table of the AMG hierarchy on its own (csrc/sns_policy.h: the ONE place that holds its size thresholds and sweep
 * schedules): for a hierarchy of `nlevels` levels with rows_global[l] rows (a replicated level: its rows) on `nranks` ranks --
 * rep_level = first level held redundantly by every rank (0: none; the level before it is only the source of the copy and gets
 * kind -1), windows != 0: a window transport (peer / team), rows_global_l1 = rows of level 1 as coarsened (decides the
 * unstructured tier of amg_nu_scale_with_size), has_blocks[l] (NULL: all 1) = the level's aggregates fit the 32 x 32 smoother
 * blocks -- the smoother / solver kind (SNS_LEVEL_*), the sweeps before and after the coarse-grid correction and whether the
 * level's sweeps are the exact global ones (amg_exact_sweeps).  What sns_get_cycle reports of a handle of that shape.            */
SNS_API int sns_host_cycle_policy(const sns_options* opt, int nranks, int windows, int nlevels, const int64_t* rows_global,
                                  int rep_level, int64_t rows_global_l1, const uint8_t* has_blocks, int32_t* kind,
                                  int32_t* nu_pre, int32_t* nu_post, int32_t* exact);

/* eigenvalues of a small real upper-Hessenberg matrix (row-major n x n, n <= 32; entries below the first subdiagonal ignored):
 * the Ritz values of the short Arnoldi process that caps the smoother damping (amg_ritz_limit).                          */
SNS_API int sns_host_hessenberg_eigs(int n, const double* H, double* re_out, double* im_out);

#ifdef __cplusplus
}
#endif
#endif /* SNS_H */
