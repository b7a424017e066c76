/* mpx.h -- C ABI of libmpx: MI355X-native pseudo-spectral collocation assembly.
 *
 * This is the drop-in boundary for ONE path of mpopt (reference: /root/reference/mpopt/mpopt.py):
 * the NLP oracle functions that `ca.nlpsol("solver","ipopt",{"f","x","g","p"},opts)` derives
 * from mpopt's transcription and that IPOPT calls every iteration
 *   nlp_f, nlp_g, nlp_grad_f, nlp_jac_g, nlp_hess_l      (mpopt.py:757, called at mpopt.py:804)
 * plus the collocation tables they are built from
 *   CollocationRoots (mpopt.py:4134-4276), Collocation (mpopt.py:3706-4131).
 * The reference has no FFI seam of its own (it is pure Python over CasADi); the entry points
 * below are what a ctypes binding replacing `ca.nlpsol`'s oracle evaluation binds to
 * (see INTEGRATION.md for the reference-side stub).
 *
 * Conventions: plain C, caller owns every buffer, no pointer is retained after a call returns
 * (device buffers belong to the context).  Every function returns 0 on success or a negative
 * MPX_ERR_* code; the message is available from mpx_last_error().  Nothing throws or aborts
 * across the ABI.  A context is used by one thread at a time.  All floating data is FP64, all
 * index data is int32 (patterns) / int64 (sizes, strides).
 *
 * Layouts (identical to the reference):
 *   z   per phase [vec(X) ; vec(U) ; t0 ; tf ; A], vec = column major (state-major):
 *       z[a*N+i] = X[i,a]; phases concatenated                       (mpopt.py:537-543, 627)
 *   p   segment widths, phase-major: p[ph*S + s]                      (mpopt.py:152, 631)
 *   g   per phase [F ; C ; DU ; mU ; dU ; TC], then the three event blocks
 *                                                                     (mpopt.py:458, 617-621)
 *   jac_g   COO triplets in a fixed library-defined order (mpx_pattern_jac), structural
 *           non-zeros only, values array matches that order; mpx_ccs_perm() gives the
 *           permutation to CasADi's compressed-column order
 *   hess_l  upper triangle (row <= col) of  sigma*f + lam_g^T g  in COO, like CasADi's
 *           "triu:hess:gamma:x:x"
 */
#ifndef MPX_H
#define MPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPX_VERSION 1

/* error codes */
#define MPX_OK 0
#define MPX_ERR_INVALID (-1)     /* bad argument / malformed structure */
#define MPX_ERR_HIP (-2)         /* a HIP runtime call failed */
#define MPX_ERR_NO_DEVICE (-3)   /* context was created without device code (structure only) */
#define MPX_ERR_UNSUPPORTED (-4)
#define MPX_ERR_ALLOC (-5)

/* collocation schemes (mpopt.py:4166-4188) */
#define MPX_SCHEME_LGR 0 /* {-1} U roots(P^(1,0)_{p-1}) U {+1}   mpopt.py:4208-4231 */
#define MPX_SCHEME_LGL 1 /* {-1} U roots(P^(1,1)_{p-1}) U {+1}   mpopt.py:4234-4259 */
#define MPX_SCHEME_CGL 2 /* cos(pi j/p) reversed                  mpopt.py:4262-4276 */
#define MPX_SCHEME_LG 3  /* {-1} U leggauss(p-1): p nodes only; tables only, no NLP */
#define MPX_SCHEME_EQUI 4 /* unknown scheme string -> linspace    mpopt.py:4182-4188 */

/* what_mask bits for mpx_eval */
#define MPX_F 1      /* nlp_f      */
#define MPX_G 2      /* nlp_g      */
#define MPX_GRAD 4   /* nlp_grad_f */
#define MPX_JAC 8    /* nlp_jac_g  */
#define MPX_HESS 16  /* nlp_hess_l */
#define MPX_JAC_VARIABLE_ONLY 64 /* opt-in, with MPX_JAC: rewrite only the jac_g entries that depend on (z, p).
                                   The differentiation / interpolation blocks are constants of the grid; a caller
                                   that keeps its jac_val buffers resident (a device-side consumer) pays for them
                                   once: the buffers must hold the values of an earlier full MPX_JAC evaluation of
                                   the same context and batch slots.  Never the default. */
#define MPX_CCS_ORDER 128 /* jac_val / hess_val leave in compressed-column order (the order mpx_ccs_perm describes,
                             CasADi's convention) instead of the native order of mpx_pattern_*: a device-side
                             permutation pass after the evaluation (~3 us), so that the nlp_jac_g / nlp_hess_l
                             entry points need no host-side reordering.  Not combinable with
                             MPX_JAC_VARIABLE_ONLY / MPX_BOUNDARY_ONLY. */
#define MPX_WIDTHS_UNCHANGED 256 /* mpx_eval_device only: `p`, p_per_point and batch are those of the previous device-pointer call
                                    of this context (mpx_eval_device or mpx_resid_eval_device) and the memory behind `p` has not
                                    changed since: the prefix sums kept on the device are reused.  The caller's assertion.
                                    (Contexts of problems none of whose node functions uses the time t never form these sums --
                                    the node time is their only consumer -- and ignore the flag.) */
#define MPX_MID_RESID 1024 /* mpx_eval_device, with MPX_HESS: the node kernels of the hess_l pass also evaluate the dynamics residuals
                              D_mid.X - h_s Sx dyn(I_mid.X, I_mid.U, t_mid, a) at the mid-points between consecutive nodes of every
                              segment -- what mpx_resid_eval_device(resid) gives for the plan whose targets are those mid-points
                              (the samples the h-adaptive loop refines on, mpopt.py:2620-2633) -- into the array registered
                              with mpx_set_mid_resid_output: one pass over z instead of two.  Degrees <= 12. */
#define MPX_BOUNDARY_ONLY 32 /* skip the node kernels: finish reductions / terminal / linking rows only
                                (second half of a segment-sharded evaluation, see mpx_set_tile_range) */
#define MPX_OWNER_RESIDENT 512 /* segment-sharded contexts only (mpx_shard_setup, world > 1): the owner-resident protocol --
                                  every rank keeps the g / grad_f rows and the jac_val / hess_val blocks of ITS tiles where
                                  they are (direct stores, no staging block) and only the per-tile partial sums are exchanged;
                                  accepted by mpx_eval_device, mpx_shard_info / _pack / _unpack (see mpx_shard_setup below) */

/* structure kinds (packed description produced by the host-side tracer) */
#define MPX_COL_X 0
#define MPX_COL_U 1
#define MPX_COL_T0 2
#define MPX_COL_TF 3
#define MPX_COL_A 4
#define MPX_ROW_F 0
#define MPX_ROW_C 1
#define MPX_TV_XF 0
#define MPX_TV_TF 1
#define MPX_TV_X0 2
#define MPX_TV_T0 3
#define MPX_TV_A 4

/* ---------------------------------------------------------------------------------------------
 * Collocation tables (host, no GPU needed).  Replace CollocationRoots._taus_fn and
 * Collocation.get_diff_matrix / get_quadrature_weights / get_interpolation_matrix
 * (mpopt.py:3815-3905, 4158-4276) with the D_MATRIX_METHOD="numerical" semantics.
 * ------------------------------------------------------------------------------------------- */

/* number of nodes of `scheme` at degree `deg` (deg+1, except LG: deg; deg==0: 1) */
int mpx_colloc_n_nodes(int scheme, int deg);
/* roots[n_nodes], ascending, mapped affinely to [tau_min, tau_max] (mpopt.py:4224) */
int mpx_colloc_roots(int scheme, int deg, double tau_min, double tau_max, double* roots);
/* D[i*n_nodes+j] = l_j^(order)(taus[i]) for the Lagrange basis on `nodes` (mpopt.py:3815-3849);
 * taus==NULL evaluates at the nodes themselves.  order is 1 or 2. */
int mpx_colloc_diff_matrix(const double* nodes, int n_nodes, const double* taus, int n_taus, int order, double* D);
/* w[j] = integral_{a}^{b} l_j(tau) dtau (mpopt.py:3851-3882) */
int mpx_colloc_quad_weights(const double* nodes, int n_nodes, double a, double b, double* w);
/* C[i*n_nodes+j] = l_j(taus[i]) (mpopt.py:3884-3905) */
int mpx_colloc_interp_matrix(const double* nodes, int n_nodes, const double* taus, int n_taus, double* C);

/* ---------------------------------------------------------------------------------------------
 * NLP context
 * ------------------------------------------------------------------------------------------- */
typedef struct mpx_ctx mpx_ctx;

typedef struct mpx_problem {
  int32_t version;          /* MPX_VERSION */
  int32_t n_phases, nx, nu, na;
  int32_t n_segments;       /* per phase (all phases share one grid, mpopt.py:70-75) */
  const int32_t* poly_orders; /* [n_segments] */
  int32_t scheme;           /* MPX_SCHEME_* */
  double tau0, tau1;        /* CollocationRoots._TAU_MIN/_TAU_MAX (mpopt.py:4144-4145) */
  /* phase links (mpopt.py:3442): n_links pairs (i, j) */
  int32_t n_links;
  const int32_t* links;
  /* packed structure, per phase in order:
   *   nc, n_tc, diff_u, midu_rows, du_continuity,
   *   NJV, NJV x (row_kind,row_comp,col_kind,col_comp),   variable Jacobian entries of a node
   *   NHN, NHN x (kind1,comp1,kind2,comp2),               node Hessian entries (upper, node row)
   *   NHC, NHC x (kind1,comp1,kind2,comp2),               (t0,tf,A) corner entries (reduced)
   *   NMG, NMG x (tv_kind,comp),                          Mayer gradient entries
   *   NTJ, NTJ x (tc_row,tv_kind,comp),                   terminal-constraint Jacobian entries
   *   NTH, NTH x (tv_kind1,comp1,tv_kind2,comp2)          terminal Hessian entries            */
  const int32_t* structure;
  int64_t structure_len;
  /* gfx950 code object holding the problem's kernels (hipcc --genco of the generated source);
   * NULL creates a structure-only context: sizes/patterns/tables work, mpx_eval fails loudly. */
  const void* code_object;
  size_t code_object_size;
  int32_t device;           /* HIP device ordinal */
} mpx_problem;

typedef struct mpx_sizes {
  int64_t n_z, n_p, n_g, nnz_jac, nnz_hess;
  int64_t n_nodes;          /* N = sum(poly_orders)+1 per phase */
  int64_t n_tiles;          /* workgroup tiles per evaluation point */
  /* algorithmic bytes per evaluation point */
  int64_t bytes_fgj;        /* 8*(2 n_z + n_p + n_g + nnz_jac + 1) */
  int64_t bytes_hess;       /* 8*(n_z + n_p + n_g + 1 + nnz_hess) */
} mpx_sizes;

int mpx_create(const mpx_problem* prob, mpx_ctx** out);
int mpx_destroy(mpx_ctx* ctx);
const char* mpx_last_error(const mpx_ctx* ctx); /* ctx==NULL: error of the last failed mpx_create */
int mpx_get_sizes(const mpx_ctx* ctx, mpx_sizes* out);

/* fixed COO patterns, 0-based (rows of g / indices of z) */
int mpx_pattern_jac(const mpx_ctx* ctx, int32_t* row, int32_t* col);
int mpx_pattern_hess(const mpx_ctx* ctx, int32_t* row, int32_t* col);
/* is_variable[nnz_jac] (order of mpx_pattern_jac): 1 where the value depends on (z, p), 0 where it is a constant of the grid --
 * copies of the differentiation / mid-point interpolation tables and the +-1 / slope coefficients of the linear rows (mpopt.py:227-232,
 * 350-369, 398-411, 484-519; about three quarters of jac_g at 1000 x 5).  What MPX_JAC_VARIABLE_ONLY rewrites is a superset of the 1s. */
int mpx_pattern_jac_variable(const mpx_ctx* ctx, uint8_t* is_variable);
/* perm[k] = position in the library's value order of the k-th entry in compressed-column order
 * (sorted by column, then row), colind[n_cols+1]; which = MPX_JAC or MPX_HESS */
int mpx_ccs_perm(const mpx_ctx* ctx, int which, int64_t* perm, int64_t* colind);

/* composite tables of the context's grid (parity hooks for SURVEY a9-a11):
 * compW[N] (mpopt.py:4041-4064), node_tau[N] reference-interval position of every node */
int mpx_get_comp_weights(const mpx_ctx* ctx, double* compW);

/* Large device-pointer batches: the library picks the launch geometry (evaluation points per workgroup) per output array by
 * timing four of the first six passes that write it (two unmeasured ones first; HIP events on the context's stream, no synchronisation; results never depend on
 * the geometry), because which geometry streams best depends on where the driver placed the array physically.  A caller that
 * re-allocates its arrays (the same address may come back on other pages) can void the measurements here; they are also
 * repeated every 512 passes. */
int mpx_geometry_reset(mpx_ctx* ctx);

/* The A/B and test switches of the EVALUATION path (environment variables MPX_NO_LIGHT, MPX_BPB, MPX_NO_LANES, ...: DESIGN.md
 * section 4's table) are read ONCE per process, at the first evaluation -- no getenv() on the call path.  on = 1: read them at every
 * call again (the test suite and the A/B tools switch them inside one process; the same as MPX_ENV_DYNAMIC=1 in the environment when
 * the library is first used); on = 0: take a new snapshot now and keep it.  Call with no evaluation in flight.  Switches read at context
 * creation are not affected. */
int mpx_env_dynamic(int on);
/* What the evaluation path sees for one of those switches (NULL: unset, or not a switch of the evaluation path). */
const char* mpx_env_knob(const char* name);

/* Use `stream` (a hipStream_t) for all subsequent work of this context; NULL = default stream. */
int mpx_set_stream(mpx_ctx* ctx, void* stream);

/* Evaluate `batch` points.  Host-pointer variant: copies in, runs, copies out, synchronous.
 *   z        [batch][n_z]
 *   p        [n_p] segment widths shared by the batch (p_per_point==0) or [batch][n_p]
 *   lam_g    [batch][n_g], sigma [batch]   (only for MPX_HESS)
 *   f [batch], g [batch][n_g], grad_f [batch][n_z], jac_val [batch][nnz_jac],
 *   hess_val [batch][nnz_hess]; outputs not selected by what_mask may be NULL.            */
int mpx_eval(mpx_ctx* ctx, int what_mask, int64_t batch, const double* z, const double* p, int p_per_point,
             const double* lam_g, const double* sigma, double* f, double* g, double* grad_f, double* jac_val,
             double* hess_val);
/* Device-pointer variant: same arguments, all pointers are device pointers, asynchronous on the
 * context's stream, nothing is copied.  This is the throughput path (inputs resident in HBM).
 * Launches per pass: one node launch per (phase, degree) bucket + the boundary pass (+ the prefix sums of the widths for problems
 * whose functions use time); on single-degree grids with several phases ALL phases share one node launch (the reference's phase
 * loop, mpopt.py:600-627, as a dimension of the grid; same results bit for bit -- MPX_NO_PHASE_MERGE=1 in the environment, read per
 * call, restores one launch per phase). */
int mpx_eval_device(mpx_ctx* ctx, int what_mask, int64_t batch, const double* z, const double* p, int p_per_point,
                    const double* lam_g, const double* sigma, double* f, double* g, double* grad_f,
                    double* jac_val, double* hess_val);
/* Block until the context's stream is idle. */
int mpx_sync(mpx_ctx* ctx);

/* nlp_grad, the sixth oracle ca.nlpsol derives from mpopt's NLP (mpopt.py:757; "nlp_grad ... n_eval 1" in every recorded solve,
 * docs/source/notebooks/moon_lander.ipynb:206): with gamma = sigma * f + lam_g^T g
 *     grad_gamma_x [batch][n_z] = sigma * grad_f + jac_g^T lam_g
 *     grad_gamma_p [batch][n_p] = d gamma / d p      (p = the segment widths: through h_s = (tf - t0) w_s / (tau1 - tau0) of the
 *                                                     segment's own nodes and through the time of every later node, mpopt.py:184-198)
 * CasADi calls it once after the last iterate with lam_f = 1 and the final multipliers; lam_p of the solver's result
 * (tests/test_examples.py:44-45) is -grad_gamma_p.  Either output may be NULL (not computed).  One fused node pass (no Jacobian is
 * stored; the transposed D / interpolation contractions run out of LDS) + one finishing pass, fixed-order sums: results do not
 * depend on the batch split.  Assembled contexts (n_p = 0) form grad_gamma_x from their gather pass.  Host-pointer variant:
 * synchronous, copies in and out; device-pointer variant: asynchronous on the context's stream.  Not available in segment-sharded
 * mode or with a tile sub-range. */
int mpx_eval_grad_gamma(mpx_ctx* ctx, int64_t batch, const double* z, const double* p, int p_per_point, const double* lam_g,
                        const double* sigma, double* grad_gamma_x, double* grad_gamma_p);
int mpx_eval_grad_gamma_device(mpx_ctx* ctx, int64_t batch, const double* z, const double* p, int p_per_point, const double* lam_g,
                               const double* sigma, double* grad_gamma_x, double* grad_gamma_p);

/* Page-locked host memory for the buffers handed to mpx_eval / mpx_resid_eval: with pageable memory
 * every transfer is staged by the runtime (~2x the latency of a single evaluation); buffers from
 * mpx_host_alloc are DMA targets.  Free with mpx_host_free before mpx_destroy. */
int mpx_host_alloc(mpx_ctx* ctx, size_t bytes, void** ptr);
int mpx_host_free(mpx_ctx* ctx, void* ptr);
/* Page-lock caller-owned memory in place (for buffers the caller cannot allocate through mpx_host_alloc, e.g. a
 * solver's work vectors).  Unregister before the memory is freed. */
int mpx_host_register(mpx_ctx* ctx, void* ptr, size_t bytes);
int mpx_host_unregister(mpx_ctx* ctx, void* ptr);

/* Segment sharding (multi-GPU): restrict the node kernels of this context to the
 * tiles [tile_begin, tile_end) and report the contiguous value ranges they own, so that ranks
 * can all-gather disjoint slices.  Default: all tiles.  The boundary kernel (reductions,
 * terminal and event rows) runs only when `run_boundary` is non-zero. */
int mpx_set_tile_range(mpx_ctx* ctx, int64_t tile_begin, int64_t tile_end, int run_boundary);
/* (Mixed-degree grids: the hess_l pass runs over its own node-ordered tiles -- runs of 256 consecutive nodes, every read
 * contiguous --; a tile range maps proportionally onto them, and mpx_shard_* reports their value runs and partial-sum slots.) */
int mpx_get_tile_jac_range(const mpx_ctx* ctx, int64_t tile, int64_t* begin, int64_t* end);
/* Relative cost of every tile (its Jacobian block size), for balancing tile ranges over ranks. */
int mpx_get_tile_weights(const mpx_ctx* ctx, int64_t* weights);
/* Mixed-degree phases: which tiles write complete g / grad_f row spans themselves.  Per tile: first node
 * and length of the span of phase nodes whose rows the tile stores (length 0: the tile stages its values for another tile or
 * for the unpack pass) and the number of nodes of other tiles inside that span.  All zero on single-degree grids and on grids
 * outside the limits of the scheme.  Arrays of n_tiles entries; host-only contexts answer too (the plan is host arithmetic). */
int mpx_get_tile_spans(const mpx_ctx* ctx, int32_t* span_first, int32_t* span_len, int32_t* n_foreign);
/* Decisions of the layout planner that are not errors but worth knowing, one per line ("" when there are none): e.g. a
 * mixed-degree grid whose row spans exceed the LDS of a compute unit (150 KB per workgroup incl. the kernel's own tiles) or whose
 * tiles would have to fetch more than 256 nodes of other buckets keeps the staging block + unpack pass for g / grad_f of the heavy
 * passes.  The string belongs to the context. */
const char* mpx_get_notes(const mpx_ctx* ctx);
/* Light passes (evaluations WITHOUT the Jacobian values: f, g, grad_f -- what a line search calls) of grids with exactly one high
 * degree (12 < P <= 31) and otherwise degrees <= 12 run through dedicated kernels (mpx_light_*): the D.X / D.U / C_mid.U contractions
 * of the high degree on the matrix cores (v_mfma_f64_16x16x4_f64), one wavefront per group of up to 16 high-degree segments plus the
 * low-degree segments between them, span-coalesced I/O.  g and the node entries of grad_f are bit-identical to the node kernels'; f
 * and the (t0, tf, a) entries of grad_f (sums over all nodes) are summed in another fixed order and may differ in the last place
 * from the heavy passes'.  Which kernels run never depends on the batch size.  Single-degree grids of degree <= 12 have the
 * same scheme without the matrix cores (mpx_lightlow_*): spans of 64 * CHL consecutive nodes per wavefront (CHL <= 12 from the LDS
 * budget of the span rows), rows of g / grad_f stored straight from the registers; their sums are defined per 64-node chunk (chunks of
 * a span in order, spans in order), and batches of fewer than 1024 spans run one chunk per wavefront with the same additions, so a
 * result never depends on the batch size it was computed in (MPX_LIGHT_LONG_SPANS=1: long spans always).  This query reports the plan: degree = 0 when
 * the grid has none, else the high degree (n_groups = groups of <= 16 high-degree segments, n_low_degree_nodes = nodes evaluated by
 * lanes) or the single low degree (n_groups = spans per phase, max_span_nodes = LDS row length, n_low_degree_nodes = 0); structure
 * only: works without a device.  MPX_NO_LIGHT=1 (environment; see mpx_env_dynamic) switches the light kernels off. */
/* Round 6: single-degree grids of degree 32 ... 255 have a third family, mpx_lighthigh_*: a workgroup = (segment, 16 evaluation points),
 * the contraction a matrix product with the evaluation points as one dimension (the transposed tables as operands from L2); the
 * query then reports degree = the grid's degree, n_groups = segments per phase, max_span_nodes = the padded K length.  Polynomial
 * degrees: every degree 1 ... 255 is accepted (round 6; degrees above MPX_TABLES_STREAM_ABOVE = 68 stream their tables instead of
 * keeping them in LDS -- up to round 5 mpx_create refused degrees >= 94). */
int mpx_get_light_plan(const mpx_ctx* ctx, int32_t* degree, int64_t* n_groups, int64_t* max_span_nodes, int64_t* n_low_degree_nodes);

/* Device buffer holding the per-tile partial sums of the last mpx_eval_device call
 * ([batch][n_tiles][width] doubles; entries of tiles outside the tile range are untouched -- except that the hess_l pass of a
 * MIXED-DEGREE grid writes the slots of its own node-ordered tiles, phase tile_first + k for its k-th tile of the phase: a tile
 * range maps proportionally onto them, so a rank's slots there need not lie inside its [tile_begin, tile_end), and the slots past a
 * phase's last node-ordered tile are nobody's; a LIGHT pass, see mpx_get_light_plan, writes one slot per group / span, phase
 * tile_first + index, its boundary pass sums exactly those, and the phase's other slots keep whatever they held).  Ranks exchange the slots they own (mpx_shard_table, kind 2) before the
 * MPX_BOUNDARY_ONLY pass. */
int mpx_get_partials(mpx_ctx* ctx, int64_t batch, double** device_ptr, int64_t* count);

/* Segment-sharded evaluation over `world` ranks (one process per GPU; the collective itself is the caller's: RCCL through
 * torch.distributed, mpopt_amd/distributed.py).  mpx_shard_setup partitions the tiles into contiguous ranges balanced by
 * Jacobian block size and puts the context into sharded mode (world == 1 leaves it):
 *     1. mpx_eval_device(mask)                       node kernels of this rank's tiles only; g / grad_f go to the packed
 *                                                    tile-ordered staging block of the context, not to the caller's arrays
 *     2. mpx_shard_pack(mask, vals, send)            this rank's owned runs -> send[rank_len * batch]
 *     3. all-gather of the send buffers              -> recv[world][rank_len * batch]               (caller, RCCL / gloo)
 *     4. mpx_shard_unpack(mask, recv, vals)          the other ranks' runs -> jac_val / hess_val, staging block, tile partials
 *     5. mpx_eval_device(mask | MPX_BOUNDARY_ONLY)   staging block -> g / grad_f; reductions, terminal and event rows
 * after which every rank holds the complete result, bit-identical to the unsharded evaluation (every entry is produced by
 * exactly one rank; reductions keep their fixed order).  `mask` selects the pass: with MPX_HESS the hess_l pass (vals =
 * hess_val), otherwise the f/g/grad_f/jac_g pass (vals = jac_val, may be NULL without MPX_JAC); run the two passes one after
 * the other.  mpx_shard_info: rank_len = padded per-rank, per-point length of the exchange buffer in doubles, n_entries =
 * rows of mpx_shard_table, tile_cuts[world + 1] = the tile ranges.  mpx_shard_table: out[n_entries][6] = (rank, kind, offset,
 * length, stride, packed_offset) of every owned run; kind 0 = jac_val / hess_val, 1 = staging block, 2 = tile partials; the
 * run of evaluation point b starts at offset + b * stride in its array and at packed_offset * batch + b * length in the
 * rank's exchange buffer (structure only: works without a device).  With MPX_OWNER_RESIDENT in `mask` the packed_offset of a kind-2
 * run is its offset in the partials-only exchange buffer of that mode (the other kinds do not travel there).
 *
 * Three ways to finish a sharded evaluation (mpopt_amd/distributed.py drives them):
 *   all-gather        steps 1-5 above: every rank ends with the complete result.
 *   gather-to-root    the same packed runs, collected by ONE rank (dist.gather): only the root runs steps 4-5 and holds the
 *                     complete result; the other ranks send rank_len * batch doubles and receive nothing.
 *   owner-resident    every call of steps 1, 2, 4, 5 carries MPX_OWNER_RESIDENT: the node kernels store g / grad_f rows directly
 *                     (no staging), pack / unpack move ONLY the tile partials (rank_len = max tiles per rank * nred doubles per
 *                     point, a few KB), and after step 5 a rank holds: f; its own node rows of g, entries of grad_f and value
 *                     blocks of jac_val / hess_val (mpx_shard_owned lists them as (offset, length) runs); and, replicated on
 *                     every rank, everything the boundary pass writes (terminal rows, control-slope continuity and event rows
 *                     with their Jacobian entries, the (t0, tf, a) entries of grad_f, the (t0, tf, a) corner of hess_l).
 *                     Entries owned by other ranks are left as they were.  A distributed consumer indexes the owner's arrays.
 */
int mpx_shard_setup(mpx_ctx* ctx, int world, int rank);
/* Owner-resident mode: the runs of output array `which` (MPX_G: rows of g, MPX_GRAD: entries of grad_f, MPX_JAC: jac_val,
 * MPX_HESS: hess_val) that `rank` owns after an evaluation, as sorted, disjoint (offset, length) pairs in runs[2 * n_runs]
 * (runs == NULL: count only).  Everything no rank owns is written by the boundary pass on every rank.  Structure only. */
int mpx_shard_owned(const mpx_ctx* ctx, int which, int rank, int64_t* n_runs, int64_t* runs);
int mpx_shard_info(const mpx_ctx* ctx, int mask, int64_t* rank_len, int64_t* n_entries, int64_t* tile_cuts);
int mpx_shard_table(const mpx_ctx* ctx, int mask, int64_t* out);
int mpx_shard_pack(mpx_ctx* ctx, int mask, int64_t batch, const double* vals, double* send);
int mpx_shard_unpack(mpx_ctx* ctx, int mask, int64_t batch, const double* recv, double* vals);

/* ---------------------------------------------------------------------------------------------
 * Off-node evaluation ("next" row): what mpopt.interpolate_single_phase and
 * mpopt.get_dynamics_residuals_single_phase compute after a solve (mpopt.py:1428-1543), batched.
 * A plan fixes the target points of one phase: points of segment s are
 * taus[seg_ptr[s] .. seg_ptr[s+1]) on the reference interval [tau0, tau1]; segments may be empty.
 * Outputs, all optional, row-major per evaluation point:
 *   ti [n_pts]       unscaled time                         (get_interpolated_time_grid, 1545-1573)
 *   xi [n_pts][nx], ui [n_pts][nu]    C.X, C.U  (scaled variables)        (1523-1532)
 *   dxi, dui         D_at.X, D_at.U
 *   dyn [n_pts][nx]  h_s * scale_x * dynamics(xi/scale_x, ui/scale_u, ti, a/scale_a)   (1466-1480)
 *   resid            dxi - dyn                                                          (1481)
 * ------------------------------------------------------------------------------------------- */
typedef struct mpx_resid_plan mpx_resid_plan;
int mpx_resid_plan_create(mpx_ctx* ctx, int phase, const int64_t* seg_ptr, const double* taus, mpx_resid_plan** out);
/* Same, with the derivative order of the D_at rows: 1 (as above) or 2, in which case dxi / dui hold the SECOND
 * derivatives of the interpolating polynomials (mpopt.get_state_second_derivative_single_phase, mpopt.py:1285-1358)
 * and dyn / resid are not meaningful. */
int mpx_resid_plan_create_order(mpx_ctx* ctx, int phase, const int64_t* seg_ptr, const double* taus, int deriv_order,
                                mpx_resid_plan** out);
int mpx_resid_plan_destroy(mpx_resid_plan* plan);
int mpx_resid_eval(mpx_ctx* ctx, mpx_resid_plan* plan, int64_t batch, const double* z, const double* p, int p_per_point,
                   double* ti, double* xi, double* ui, double* dxi, double* dui, double* dyn, double* resid);
int mpx_resid_eval_device(mpx_ctx* ctx, mpx_resid_plan* plan, int64_t batch, const double* z, const double* p,
                          int p_per_point, double* ti, double* xi, double* ui, double* dxi, double* dui, double* dyn,
                          double* resid);

/* Output of the MPX_MID_RESID passes: device array [batch][n_phases * (N - 1)][nx]; row ph * (N - 1) + i - 1 = the mid-point
 * between nodes i - 1 and i of phase ph (segment order, the order of a residual plan over the mid-points).  NULL switches off. */
int mpx_set_mid_resid_output(mpx_ctx* ctx, double* resid);

/* Width update of the h-adaptive refinement loop on the device, batched: the equal-area rule
 * mpopt_h_adaptive.get_roots_wrt_equal_area (mpopt.py:2636-2659) applied to r_i = || resid[b][i][0..nx) ||_2, i < n_pts (the
 * residual samples of the phase in segment order, e.g. the `resid` output of mpx_resid_eval_device; mpopt.py:2620-2633),
 * followed by the reference's damped update (mpopt.py:2587-2590):
 *     p_out[b][phase*S + s] = damping * new_width_s + (1 - damping) * p_in[(b)][phase*S + s]        (the reference: 0.4)
 * p_in is [n_p] (p_in_per_point == 0) or [batch][n_p]; p_out is [batch][n_p]; other phases' entries are not touched.
 * Device pointers, asynchronous on the context's stream.  For n_pts <= 12288 samples and <= 4096 segments per phase the update also
 * leaves the exclusive prefix sums of the phase's new widths on the device (what the next evaluation would compute from p_out, same
 * additions in the same order): once every phase has been updated, mpx_eval_device(..., p = p_out, p_per_point = 1, same batch) may
 * pass MPX_WIDTHS_UNCHANGED. */
int mpx_equal_area_widths_device(mpx_ctx* ctx, int phase, int64_t batch, int64_t n_pts, const double* resid, const double* p_in,
                                 int p_in_per_point, double* p_out, double damping);

/* ---------------------------------------------------------------------------------------------
 * Assembled contexts: transcriptions whose NLP has the form
 *
 *     f(z) = F . phi(L z),      g(z) = G_z z + g_0 + G . phi(L z)
 *
 * with phi a set of generated point functions (collocation nodes, mid-points, phase ends) applied to
 * local variables that are LINEAR in z, and constant sparse maps around them.  Used for
 * mpopt_adaptive (reference mpopt.py:2877-3375: segment widths are decision variables, mid-point
 * residual rows couple all nodes of a segment).  Evaluation is two-stage and reduction-order fixed:
 *   (1) point kernels (generated code, lane <-> point) write the point values / their structural
 *       first or second derivatives to a raw buffer;
 *   (2) one gather kernel forms every output entry as a fixed-order sum  sum_k coef_k * v[src_k]
 *       over  v = [raw ; z ; 1]  (the products  G * dphi * L  and  L^T * d2phi * L  are expanded once,
 *       on the host, into these rows).
 * The context answers mpx_get_sizes / mpx_pattern_* / mpx_ccs_perm / mpx_eval / mpx_eval_device /
 * the nlp_* symbols like any other; it has no parameters (n_p = 0, `p` may be NULL) and no tiles.
 * ------------------------------------------------------------------------------------------- */
typedef struct mpx_point_set {
  int32_t fid;              /* generated function mpxgen::Pt<fid> of the code object (kernels mpx_pts_val / _jac / _hes) */
  int32_t n_points;
  int32_t n_loc, n_cst, n_out, n_jac, n_hess; /* sizes of the generated function (checked against nothing: caller's contract) */
  /* local variable v of point p:  sum_{t < loc_nterm[v]} loc_coef[(toff_v + t) * n_points + p] * z[loc_idx[...same...]] */
  const int32_t* loc_nterm; /* [n_loc] */
  const int32_t* loc_idx;
  const double* loc_coef;
  const double* cst;        /* [n_cst][n_points] per-point constants */
  /* Hessian multiplier of output r of point p: same layout over [lam_g ; sigma] (index n_g = sigma) */
  const int32_t* mu_nterm;  /* [n_out] */
  const int32_t* mu_idx;
  const double* mu_coef;
} mpx_point_set;

/* rows of sums over v = [raw ; z ; 1]:  src >= 0 raw slot, src == -1 the constant 1, src <= -2 z[-2 - src] */
typedef struct mpx_gather {
  int64_t n_rows;
  const int64_t* ptr;   /* [n_rows + 1] */
  const int32_t* src;
  const double* coef;
} mpx_gather;

typedef struct mpx_assembly {
  int32_t version;      /* MPX_VERSION */
  int64_t n_z, n_g, nnz_jac, nnz_hess;
  int32_t n_sets;
  const mpx_point_set* sets;
  /* raw slot of (set k, slot q, point p) = raw_off_k + q * n_points_k + p, raw_off_k = running sum of
   * n_points * (n_out + n_jac) for first-order passes (slots: outputs, then Jacobian entries) and of
   * n_points * n_hess for the Hessian pass */
  mpx_gather fgj;       /* rows in order: f, g[n_g], grad_f[n_z], jac_val[nnz_jac] */
  mpx_gather hess;      /* rows: hess_val[nnz_hess] */
  const int32_t *jac_row, *jac_col, *hess_row, *hess_col;
  const void* code_object;
  size_t code_object_size;
  int32_t device;
} mpx_assembly;

int mpx_create_assembled(const mpx_assembly* desc, mpx_ctx** out);
/* Which batched kernels the code object of an assembled context carries (0: none; single evaluations and small batches always
 * run the two-pass kernels): fused_lanes = lanes per workgroup of the fused persistent kernels (batches of >= 256 points);
 * hess_lane_groups / first_order_lane_groups = groups of point tasks of the lane-per-evaluation-point kernels (mpx_asml_hes: hess_l;
 * mpx_asml_fgj: f + g + grad_f + jac_g requested together, an opt-in of the generator; batches of >= 64 / 512 points; one wavefront per group and 64 evaluation
 * points, a second small kernel for rows that sum over all groups).  All paths give the same bits. */
int mpx_get_assembled_plan(const mpx_ctx* ctx, int32_t* fused_lanes, int32_t* hess_lane_groups, int32_t* first_order_lane_groups);
/* Attach a second code object with the lane-per-evaluation-point kernels (mpx_asml_*, generated by mpopt_amd/assembly_lanes.py for
 * the same transcription and the same order of the patterns) to an assembled context.  They serve batches only, and their straight-line
 * code takes as long to compile as everything else of the context together: a caller that only ever evaluates single points (IPOPT)
 * never pays for them -- the Python layer attaches them at the first batch of >= 64 points.  The image is handed to hipModuleLoadData
 * like mpx_assembly.code_object; the Python layer keeps both buffers for the life of the context. */
int mpx_assembled_attach_kernels(mpx_ctx* ctx, const void* code_object, size_t code_object_size);

/* ---------------------------------------------------------------------------------------------
 * CasADi-external-compatible surface (mpx_casadi.cpp): the symbols nlp (the base oracle (x, p) -> (f, g) that
 * ca.nlpsol(name, solver, "libmpx.so") resolves first, as external("nlp", ...)), nlp_f, nlp_g, nlp_grad_f, nlp_jac_g, nlp_hess_l and
 * nlp_grad (+ _n_in/_n_out/_name_in/_name_out/_sparsity_in/_sparsity_out/_work/_incref/_decref, and the optional
 * _alloc_mem/_init_mem/_free_mem/_checkout/_release/_default_in) follow the calling convention of CasADi-generated C code and act on
 * the context selected here (process-wide; NULL clears it).  The context must outlive its selection.
 * ------------------------------------------------------------------------------------------- */
int mpx_set_current(mpx_ctx* ctx);
/* Opt-in: page-lock the argument / result arrays handed to nlp_* the first time each pointer is seen (CasADi passes
 * the same work-vector slices on every call), so that every transfer is a direct DMA.  The registrations are
 * released by mpx_set_current (any argument) and by mpx_current_pin_buffers(0); the arrays must outlive that. */
int mpx_current_pin_buffers(int enable);
/* Same-iterate coalescing of the nlp_* entry points (mpx_casadi.cpp): the first call at a new (x, p) evaluates f, g and grad_f
 * (and jac_g when nnz_jac * 8 <= 64 KB) in one fused pass; nlp_f / nlp_g / nlp_grad_f (and that small nlp_jac_g) at the same
 * point are then served from page-locked scratch of the context.  Counters since the context was selected: fused device passes
 * made by the cache and calls answered without a device pass.  MPX_NO_COALESCE=1 (environment, read by mpx_set_current)
 * switches the cache off. */
int mpx_current_cache_stats(long long* fused_passes, long long* served_from_cache);
/* The name nlp_hess_l_name_out(0) answers with.  CasADi resolves the functions of an external oracle BY NAME and compares every
 * input / output name with its request string (':' -> '_'); the request for the Hessian of the Lagrangian differs between the
 * versions the reference admits (setup.py:29 casadi >= 3.5.5; requirements.txt:4 casadi == 3.6.0):
 *   3.6.x  "triu:hess:gamma:x:x" -> triu_hess_gamma_x_x   (default; mpx_current_set_casadi_abi(306))
 *   3.5.x  "sym:hess:gamma:x:x"  -> sym_hess_gamma_x_x    (mpx_current_set_casadi_abi(305); from the 3.5.5 sources as remembered --
 *          no CasADi of any version is available to the build)
 * Both are the same upper-triangular compressed-column matrix.  mpx_current_set_hess_l_output_name sets any other identifier
 * ([A-Za-z0-9_], < 64 characters) -- e.g. the one a future importer's "Inconsistent output name. Expected: ..." message names.
 * Process-wide, like the current context; call before ca.nlpsol(...). */
int mpx_current_set_casadi_abi(int major_minor);
int mpx_current_set_hess_l_output_name(const char* name);
/* Opt-in for nlp_jac_g: leave the constants of a large Jacobian in the caller's array.  After a full pass into res[1], later calls
 * with the SAME res[1] rewrite only the (z, p)-dependent entries (a single evaluation is bound by what it writes over PCIe: 0.96 MB ->
 * 0.25 MB at 1000 x 5).  CONTRACT: nothing but nlp_jac_g writes to that array between the calls -- true of the work-vector slice
 * CasADi's nlpsol passes.  As a safety net ~500 sampled constants are compared before every partial pass (a cleared, reused or
 * reallocated array falls back to the full pass); a caller that alters single constants is not detected -- hence opt-in.
 * mpx_current_jac_stats counts partial and full passes since the context was selected. */
int mpx_current_keep_jac_constants(int enable);
int mpx_current_jac_stats(long long* variable_only_passes, long long* full_passes);
/* Caller arrays page-locked so far by mpx_current_pin_buffers(1) and registrations that failed (remembered, not retried): a
 * solver that passes the same work-vector slices on every call stops adding to these after its first iteration. */
int mpx_current_pin_stats(long long* registered, long long* failed);

/* ---------------------------------------------------------------------------------------------
 * Timing helper: HIP events on the context's stream (bench.py measures kernel time with these)
 * ------------------------------------------------------------------------------------------- */
/* PCI bus id ("0000:c1:00.0") of HIP device `device` into out[len]: lets a multi-process caller prove that its ranks sit on
 * distinct GPUs (bench.py's `rccl` object). */
int mpx_device_pci_bus_id(int device, char* out, int len);
int mpx_timer_start(mpx_ctx* ctx);
int mpx_timer_stop(mpx_ctx* ctx, double* elapsed_ms); /* records stop, synchronises, returns ms */
/* Per-kernel timing: while enabled, every mpx_eval_device brackets its node-kernel launches (the
 * dominant kernels) with a pair of HIP events on the context's stream.  mpx_profile_read
 * synchronises, returns the summed node-kernel time and the number of bracketed launches since
 * the last read, and resets the counters. */
int mpx_profile(mpx_ctx* ctx, int enable);
int mpx_profile_read(mpx_ctx* ctx, double* node_ms_total, int64_t* n_node_launches);

#ifdef __cplusplus
}
#endif
#endif /* MPX_H */
