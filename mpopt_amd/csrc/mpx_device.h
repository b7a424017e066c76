// mpx_device.h -- argument blocks shared by the host runtime (mpx_host.cpp) and the kernels
// (mpx_kernels.h, compiled per problem).  Plain-old-data only.
#ifndef MPX_DEVICE_H
#define MPX_DEVICE_H
#include <stdint.h>

#define MPX_TILE 256           // nodes (= lanes) per workgroup tile: 4 wavefronts of 64
#define MPX_MAX_PHASES 8
// Polynomial degrees above this do not keep the differentiation / mid-point tables in LDS: the node kernels stream them from the
// transposed tables in global memory (mpx_kernels.h: node_body, TAB_GLB).  A build-time constant of the code object
// (-DMPX_TABLES_STREAM_ABOVE=n; exported as `mpx_tables_stream_above`, which the host compares with its own value: the
// environment variable MPX_TABLES_STREAM_ABOVE at context creation, else this default) -- the host lays the tables out accordingly.
// 68: up to there two workgroups share a compute unit with the tables in LDS (2 * 69^2 doubles = 76 KB each) and the LDS form is
// the faster one (tools/r6_stream_ab.py, B = 256, f+g+grad_f+jac_g in TB/s, LDS | streamed: degree 48 5.0 | 4.1, 56 5.5 | 5.5,
// 64 5.6 | 5.3, 80 3.4 | 4.5, 92 3.0 | 4.7; g alone at 92: 468 | 131 us).
#ifndef MPX_TABLES_STREAM_ABOVE
#define MPX_TABLES_STREAM_ABOVE 68
#endif
#define MPX_LIGHT_WAVES 4       // wavefronts per workgroup of the light-pass kernels (mpx_light_*): two workgroups per compute unit

// kernel modes
#define MPX_MODE_FG 0    // f, g
#define MPX_MODE_FGJ 1   // f, g, grad_f, jac_g
#define MPX_MODE_HESS 2  // hess_l

// One workgroup's share of one (phase, degree) bucket: whole segments, <= MPX_TILE nodes.
struct MpxTile {
  int32_t m0;        // first bucket-local node
  int32_t n;         // lanes that load a node (stage it in LDS)
  int32_t n_own;     // lanes that own outputs (== n, except the node-0 tile: 1)
  int32_t node0;     // 1: this is the mini-tile of node 0 of the phase (lanes 1..P only feed LDS;
                     //    node 0 has no mid-point row).  Regular tiles hold whole segments, points 1..P.
  int32_t tile_id;   // slot in the partial-sum buffer, global over phases and buckets
  int32_t seg0;      // first segment of the tile
  int64_t jac_base;  // offset of the tile's block in the jac_g value array
  int64_t hess_base; // offset of the tile's block in the hess_l value array
  int64_t g_base;    // offset of the tile's block in the packed g / grad_f staging buffer (mixed-degree phases)
  // Absorbing tiles (mixed-degree phases, see MpxNodeArgs::abs_cap): the tile writes the complete g / grad_f rows of the node span
  // [span_lo, span_lo + span_len) -- its own nodes from registers, the other buckets' nodes in between (f_count of them, entries
  // f_first .. of the abs_* lists) from the staging buffer.  span_len == 0: the tile stages its values.
  int32_t span_lo, span_len, f_first, f_count;
};

// Batched I/O views: pointer + per-evaluation-point stride (in doubles).
struct MpxIO {
  const double* z;      int64_t z_stride;
  const double* w;      // segment widths   [n_wvec][n_phases*S]
  const double* wcum;   // exclusive prefix [n_wvec][n_phases*S]
  int64_t w_stride;     // 0: widths shared by the batch
  const double* lam_g;  int64_t lam_stride;
  const double* sigma;
  double* f;
  double* g;            int64_t g_stride;
  double* grad;         int64_t grad_stride;
  double* jac;          int64_t jac_stride;
  double* hess;         int64_t hess_stride;
  double* partial;      // [B][n_tiles_total][nred]
  int32_t n_tiles_total, nred;
  int32_t B, b_per_block;
  int32_t b_first, pad_b_;  // first evaluation point of this launch (batches of more than 65535 points per workgroup row are launched in slices)
  int32_t jac_variable_only, pad_;
  // Mixed-degree phases: the nodes of one (phase, degree) bucket are NOT contiguous in the node index, so direct
  // stores to g / grad_f are short runs with gaps that another kernel fills later (partial cache lines: measured
  // 4x the cost per byte on config 3).  Non-NULL: node kernels write [tile][slot][lane] here (coalesced) and
  // mpx_unpack_kernel moves the values to their rows with fully coalesced stores.
  double* gtmp;         int64_t gtmp_stride;
  // MPX_MID_RESID (with the hess_l pass): dynamics residuals at the mid-points between consecutive nodes of every segment,
  // [B][n_phases * (N - 1)][nx]; the row of the mid-point between nodes i - 1 and i of phase ph is ph * (N - 1) + i - 1.  NULL: off.
  double* mid_resid;    int64_t mid_stride;
};

// Node kernels: one launch per (phase, degree) bucket.
struct MpxNodeArgs {
  MpxIO io;
  const MpxTile* tiles;    // tiles of this bucket (blockIdx.x indexes it, offset by tile_first)
  const int32_t* node_i;   // bucket-local node -> node index in the phase
  const int32_t* node_sk;  // bucket-local node -> (segment << 8) | point
  const double* Dmat;      // (P+1)x(P+1) row-major first-derivative matrix of this degree ...
  const double* Cmid;      // P x (P+1) interpolation to the mid-points between consecutive nodes ...
                           // ... degrees above MPX_TABLES_STREAM_ABOVE: both TRANSPOSED ([j][k], [j][k - 1]), streamed by the lanes
  const double* tk;        // (tau_k - tau0)/(tau1 - tau0), k = 0..P
  const double* Dmid;      // P x (P+1): first derivative of the Lagrange basis at the mid-points (MPX_MID_RESID)
  const double* tkm;       // ((tau_{k-1} + tau_k)/2 - tau0)/(tau1 - tau0), k = 1..P at index k - 1
  int32_t phase, deg;      // (deg: the bucket's polynomial degree -- the resident kernel dispatches on (phase, deg))
  const double* Wnode;     // composite quadrature weight per node of this phase
  double inv_dtau;         // 1/(tau1 - tau0)
  int64_t z_off;           // offset of the phase's block in z / grad_f
  int64_t g_off_F, g_off_C, g_off_DU, g_off_mU;  // row offsets of the phase's blocks in g
  int32_t N, seg_off;      // nodes per phase; phase * n_segments (offset into the width vectors)
  int32_t tile_first, tile_count;  // tile sub-range to run (segment sharding)
  int32_t regular, pad_;           // 1: the bucket holds every segment of the phase (node_i[m] == m, node_sk by arithmetic)
  // regular buckets: the tile descriptors by arithmetic too (tile 0 = the node-0 mini tile, tiles 1 .. reg_last = whole segments,
  // reg_lanes nodes each, the last one shorter).  Index 0: node-0 tile, 1: first full tile (the others follow at reg_size strides),
  // 2: last tile.
  int32_t reg_first_tile, reg_last, reg_lanes, reg_last_lanes;
  int64_t reg_jac_base[3], reg_hess_base[3], reg_g_base[3];
  int64_t reg_jac_size, reg_hess_size, reg_g_size;
  // Mixed-degree phases without the unpack pass: the bucket with the most nodes is launched LAST and its tiles assemble whole
  // row spans in LDS (abs_cap doubles per row slot; dynamic shared memory) and store them as full contiguous runs.  Foreign node
  // f of a tile: position abs_fpos[f] in the span, staged values at gtmp[abs_fstage[f] + slot * abs_fn[f]].  abs_cap == 0: off.
  const int32_t* abs_fpos;
  const int64_t* abs_fstage;
  const int32_t* abs_fn;
  int32_t abs_cap, pad2_;
};

// All phases of a single-degree grid in ONE launch (mpx_node_<mode>_all_<deg>; round 5): the phases of an OCP share one grid
// (mpopt.py:70-75), so their buckets have the same degree and differ only in offsets and in the generated node functions.  A
// launch per phase pays a prologue, a tail and a kernel boundary each (config 4: two phases of 7 tiles) -- here the workgroups
// of all phases are ONE grid: item -> (tile of the launch, evaluation point), XCD-blocked over the whole grid; tile t belongs to
// the phase with tile_cum[ph] <= t < tile_cum[ph + 1] and is tile t - tile_cum[ph] of that phase's range.  The kernel-side
// array has the code object's phase count (MPX_NPH, generated source); the host passes the first n_ph entries.
#ifdef MPX_NPH
#define MPX_NPH_ARGS MPX_NPH
#else
#define MPX_NPH_ARGS MPX_MAX_PHASES
#endif
struct MpxNodeMultiArgs {
  int32_t tile_cum[MPX_MAX_PHASES + 1];
  int32_t n_ph;
  MpxNodeArgs a[MPX_NPH_ARGS];
};

// Light passes (f, g, grad_f without the Jacobian values) of a phase whose grid has ONE high degree (12 < P <= 31, contractions on the
// matrix cores) and otherwise low degrees (<= 12), mpx_kernels.h: light_body.  A wavefront works on a GROUP: up to 16 consecutive
// segments of the high degree plus every low-degree segment between them -- one contiguous span of the phase's nodes, read and
// written with fully coalesced accesses through an LDS buffer.
#define MPX_LIGHT_MAXDEG 8  // distinct degrees of a grid with a light plan
#define MPX_LIGHT_CHUNKS 10
#ifndef MPX_LOW_MAX_CHUNKS
#define MPX_LOW_MAX_CHUNKS 12  // 64-node chunks of a span of the low-degree light kernels (light_low_body; the host's plan uses the same cap)
#endif // a group's span: at most 64 * MPX_LIGHT_CHUNKS nodes (one load per lane, chunk and row, all in flight together)
struct MpxLightGroup {
  int32_t lo_r, len_r;      // nodes read: [lo_r, lo_r + len_r) (the owned span and the node before it)
  int32_t lo_w, len_w;      // nodes owned (rows written): [lo_w, lo_w + len_w)
  int32_t seg_first, n_light;  // the group's high-degree segments: bucket-local indices seg_first .. seg_first + n_light - 1
  int32_t f_first, f_count; // its foreign nodes: entries of MpxLightArgs::foreign
};
struct MpxLightForeign {    // one node of a low-degree segment inside a group's span
  int32_t pos, pos0;        // the node and point 0 of its segment, relative to lo_r
  int32_t dk;               // (index of the degree table << 8) | point k
  int32_t s;                // segment
  double tk, w;             // (tau_k - tau0) / (tau1 - tau0) and the quadrature weight of the node (no dependent table loads in the kernel)
};
struct MpxLightArgs {
  MpxNodeArgs node;         // io, table of the high degree (Dmat, Cmid, tk), offsets; node_i / node_sk of the high-degree bucket
  const MpxLightGroup* groups;
  const MpxLightForeign* foreign;
  const double* wdeg;       // [P + 1] quadrature weights of the high degree
  const double* ftab;       // differentiation and mid-point tables of the low degrees, concatenated (copied to LDS by every workgroup)
  int32_t ftab_n;           // doubles in ftab
  int32_t fD_off[MPX_LIGHT_MAXDEG], fC_off[MPX_LIGHT_MAXDEG];  // offsets in ftab, by degree-table index
  int32_t fdeg[MPX_LIGHT_MAXDEG];
  int32_t n_groups, first_node;  // first_node: bucket-local node of point 1 of the bucket's first whole segment (0 or 1)
  int32_t span_cap, slot_first;  // LDS doubles per row and wavefront; first partial-sum slot of the phase (group g writes slot_first + g)
  long long* dbg;           // MPX_LIGHT_DEBUG=1 (code objects built with -DMPX_LIGHT_STAMPS): phase stamps of one wavefront, else NULL
};

// ... and the low-degree span kernels of all phases in one launch (mpx_lightlow*_all_<deg>): item -> (span, phase, evaluation point);
// what differs between the phases of a grid is the offsets below (and the generated functions).  `base` carries phase 0.
struct MpxLightPhase {
  int64_t z_off, g_off_F, g_off_C, g_off_DU, g_off_mU;
  int32_t seg_off, slot_first;
};
struct MpxLightMultiArgs {
  MpxLightArgs base;
  int32_t n_ph, pad_;
  MpxLightPhase ph[MPX_NPH_ARGS];
};

// Mixed-degree grids, hess_l pass: the node Hessian does not depend on the polynomial degree (no D.X contraction), so its tiles
// are runs of MPX_TILE CONSECUTIVE nodes of a phase instead of the (phase, degree) buckets' tiles: every z / lam_g read is one
// contiguous run (the degree-3 bucket of config 3 fetched whole cache lines for 48-byte runs: 26 % of the pass for 1/6 of the
// nodes) and the pass is one launch per phase.
struct MpxHTile {
  int32_t i0, n;        // first node of the tile, nodes (= owning lanes)
  int32_t tile_id, pad; // slot in the partial-sum buffer
  int64_t hess_base;    // offset of the tile's block in the hess_l value array
};
struct MpxHessNodeArgs {
  MpxIO io;
  const MpxHTile* htiles;   // all phases; blockIdx.x + tile_first indexes it
  const int32_t* node_seg;  // [N] segment that owns node i (the earlier one at a shared node, mpopt.py:189-195)
  const double* node_tk;    // [N] (tau_k - tau0)/(tau1 - tau0) of node i in its segment
  const double* Wnode;      // [N] composite quadrature weight
  double inv_dtau;
  int64_t z_off, g_off_F, g_off_C;
  int32_t N, seg_off, tile_first, tile_count;
};

// Linear rows handled by the boundary kernel (control-slope continuity dU, phase-link events):
// g[row] = sum_e coef[e] * z[idx[e]], Jacobian values = coef.
struct MpxPhaseInfo {
  int64_t z_off;
  int32_t N, tile_first, tile_count, tile_count_h;  // tile_count_h: tiles of the hess_l pass (== tile_count unless its tiles are node-ordered)
  int64_t g_off_TC;      // first terminal-constraint row
  int64_t jac_TC;        // first terminal-constraint Jacobian value
};

struct MpxBoundArgs {
  MpxIO io;
  MpxPhaseInfo ph[MPX_MAX_PHASES];
  // destinations (host built): grad index for every Mayer-gradient entry, hess index for every
  // corner / terminal Hessian entry (bit 62 set: accumulate into an entry a node kernel wrote)
  const int64_t* mg_dst;   // concatenated over phases
  const int64_t* hc_dst;
  const int64_t* th_dst;
  int32_t mg_off[MPX_MAX_PHASES], hc_off[MPX_MAX_PHASES], th_off[MPX_MAX_PHASES];
  // linear rows
  const int64_t* lin_ptr;  // [n_lin+1]
  const int64_t* lin_idx;  // z index
  const double* lin_coef;
  const int64_t* lin_row;  // g row of each linear row
  int64_t lin_jac;         // first Jacobian value of the linear rows
  int32_t n_lin;
  int32_t part_group;      // > 1: partial-sum slots are added in groups of this many (short-span light passes, light_low_body)
};

// nlp_grad (the sixth oracle of ca.nlpsol, mpopt.py:757): gradient of gamma = sigma * f + lam_g^T g w.r.t. x and w.r.t. the
// parameters p (segment widths).  Node pass mpx_node_gradl_<ph>_<deg>: one launch per (phase, degree) bucket, lane <-> node,
// ONE evaluation point per workgroup; finishing pass mpx_gradl_finish: one workgroup per evaluation point.
struct MpxGradlArgs {
  const double* z;      int64_t z_stride;
  const double* w;      const double* wcum; int64_t w_stride;
  const double* lam_g;  int64_t lam_stride;
  const double* sigma;
  double* gx;           int64_t gx_stride;  // grad_gamma_x [B][n_z]; NULL: only what grad_gamma_p needs is computed
  double* halo;         // [B][n_phases * S][nx + nu]: what the rows of segment s >= 1 contribute to the columns of ITS FIRST node --
                        //   the last node of segment s - 1 -- where that node's lane is NOT the previous lane of the same workgroup (the
                        //   first segment of a tile; a neighbour of another degree): added by the finishing pass.  Everywhere else the
                        //   owner of the node adds it itself (round 6: 999 of 1000 segments at configs[1])
  double* pseg;         // [B][n_phases * S][2]: per SEGMENT the sums over its nodes of (d gamma_i / d w_s of the node's own segment,
                        //   d gamma_i / d th), added from the last node down (round 6: per node before, summed by the finishing pass)
  double* partial;      // [B][n_tiles_total][nred]: tile sums of d gamma_i / d (t0, tf, A)
  int32_t n_tiles_total, nred;
  int32_t B, b_first;
  const MpxTile* tiles;
  const int32_t* node_i;
  const int32_t* node_sk;
  const double* Dmat;
  const double* Cmid;
  const double* tk;
  const double* Wnode;
  double inv_dtau;
  int64_t z_off, g_off_F, g_off_C, g_off_DU, g_off_mU;
  int32_t N, seg_off, tile_first, phase, S;
  int32_t bpb;          // evaluation points a workgroup takes, one after the other (B = one past the last point of this launch)
};
struct MpxGradlFinArgs {
  const double* z;      int64_t z_stride;
  const double* lam_g;  int64_t lam_stride;
  const double* sigma;
  double* gx;           int64_t gx_stride;
  double* gp;           int64_t gp_stride;  // grad_gamma_p [B][n_phases * S]; NULL: not requested
  const double* halo;
  const double* pseg;
  const double* partial;
  int32_t n_tiles_total, nred;
  const int32_t* halo_seg;                  // the segments whose `halo` entry the node pass wrote, phase by phase
  int32_t halo_off[MPX_MAX_PHASES + 1];     // phase p: halo_seg[halo_off[p] .. halo_off[p + 1])
  MpxPhaseInfo ph[MPX_MAX_PHASES];
  const int32_t* seg_start;  // [S + 1]
  int32_t S, n_lt;
  // linear rows (control-slope continuity, phase events) transposed: column lt_col[c] of z receives
  // sum_{e in [lt_ptr[c], lt_ptr[c + 1])} lam_g[lt_row[e]] * lt_coef[e]
  const int64_t* lt_ptr;
  const int64_t* lt_col;
  const int64_t* lt_row;
  const double* lt_coef;
};

// Off-node evaluation (interpolated trajectories and dynamics residuals, mpopt.py:1428-1543):
// one launch per degree bucket of a residual plan; lane <-> target point.
struct MpxResidArgs {
  const double* z;      int64_t z_stride;
  const double* w;      const double* wcum; int64_t w_stride;
  const int32_t* pt_id;   // [n] row of the point in the outputs
  const int32_t* pt_seg;  // [n] segment of the point
  const double* pt_tn;    // [n] (tau - tau0)/(tau1 - tau0)
  const double* Cmat;     // [P+1][n] interpolation row of every point (transposed: lane-contiguous)
  const double* Dmat;     // [P+1][n] first-derivative row of every point
  const int32_t* seg_start;  // [S+1] first node of every segment
  double* ti;           // [B][n_pts]
  double* xi;           // [B][n_pts][nx]
  double* ui;           // [B][n_pts][nu]
  double* dxi;          // [B][n_pts][nx]
  double* dui;          // [B][n_pts][nu]
  double* dyn;          // [B][n_pts][nx]   h_s * Sx * dyn(...)
  double* resid;        // [B][n_pts][nx]   dxi - dyn
  double inv_dtau;
  int64_t z_off;
  int32_t n, n_pts, N, seg_off, B, b_per_block;
};

#define MPX_ACCUM_BIT (1LL << 62)

// ---- resident kernel (single evaluations, the regime an NLP solver drives) ------------------------------------------------
// One workgroup per tile stays on the device and is fed through a mailbox in page-locked host memory: no kernel launch, no
// stream synchronisation per evaluation.  Workgroup 0 polls `seq`; a new value is a request: it copies `mode`, `ccs` and `io` to
// device memory and publishes the number there for the other workgroups; node pass (every workgroup its tile) -> grid barrier ->
// boundary pass (workgroup 0) -> [grid barrier -> compressed-column permutation of the values, all workgroups] -> grid barrier ->
// `done = seq`.  The kernel leaves by itself when idle for idle_ticks or older than life_ticks (wall_clock64, 100 MHz): it first
// clears `alive`, then looks at `seq` once more (the host writes `seq` BEFORE it reads `alive`), and sets `exited` as its last act.
#define MPX_RES_SLOTS 8      // argument sets the device remembers: an NLP solver repeats a handful of (function, arrays) combinations
struct MpxResRequest {
  int32_t mode, ccs;         // MPX_MODE_*; ccs: bit 0 jac_val, bit 1 hess_val leave in compressed-column order
  MpxIO io;                  // B = 1; with ccs the value pointer of io is the native-order scratch and ccs_out the caller's array
  double* ccs_out;
};
struct MpxMailbox {
  // host -> device, ONE word polled over PCIe: (request number << 8) | (argument slot << 1) | (1: the slot's content is new, read it
  // from `slots`).  In the steady state of a solve a request costs the device nothing but this read.
  unsigned long long seq;
  unsigned long long done;   // device -> host: request number of the last completed request
  unsigned int alive, exited, stop, pad_;
  MpxResRequest slots[MPX_RES_SLOTS];
};
struct MpxResidentArgs {
  MpxMailbox* box;                  // device alias of the mailbox
  const MpxNodeArgs* buckets;       // static part of every bucket's arguments (io comes with the request)
  const int32_t* tile_bucket;       // [n_tiles]
  const MpxBoundArgs* bound;
  MpxResRequest* dev_slots;         // device copies of the argument slots
  unsigned long long* dev_seq;      // the request word, republished in device memory for the workgroups other than 0
  unsigned long long* sync_count;   // grid barrier: arrivals since the kernel started
  const int64_t* perm_j;            // compressed-column permutations (NULL until first needed)
  const int64_t* perm_h;
  int64_t nnz_j, nnz_h;
  unsigned long long start_seq;     // the last request served before this launch
  long long idle_ticks, life_ticks;
  int32_t n_tiles, pad_;
};

// Segment sharding: one contiguous run of values a rank owns.  kind 0: jac_val / hess_val of the caller, 1: packed g / grad_f
// staging, 2: per-tile partial sums.  The run of evaluation point b starts at  src_off + b * stride  in its array and at
// dst_off * B + b * len  in the rank's exchange buffer.
struct MpxShardEnt {
  int64_t src_off, len, stride, dst_off;
  int32_t kind, rank;
  int64_t part_off;  // kind 2: offset among the rank's partial-sum runs (the owner-resident exchange carries nothing else)
};

// ---- assembled contexts (mpx_create_assembled) ---------------------------------------------------
// One point set (device-resident array, fixed for the life of the context).
struct MpxPtSet {
  int32_t n, fid, block_first, n_slots_fgj, n_hess;  // block_first: first 64-lane block of this set in the fused launch
  const int32_t* loc_toff;  // [NLOC + 1] running term offsets
  const int32_t* loc_idx;   // [terms][n]
  const double* loc_coef;
  const double* cst;        // [NCST][n]
  const int32_t* mu_toff;   // [NOUT + 1]
  const int32_t* mu_idx;
  const double* mu_coef;
  int64_t raw_off, rawh_off;
  // Chained local variable (fused kernels): variable chain_v of every point is a PREFIX of one shared term list (the running sum of
  // the earlier segment widths, mpopt.py:186-195: 19 of the 25 / 52 table terms of a moon-lander node / mid-point).  The list is
  // summed once per evaluation point -- a sequential fma chain, i.e. exactly the partial sums every point would compute -- into
  // the chain slots of LDS, and point p reads slot chain_pos[p].  chain_v < 0: none.
  int32_t chain_v, n_loc;
  const int32_t* chain_pos;  // [n]
  int32_t n_out, pad_;       // (n_loc / n_out: lengths - 1 of loc_toff / mu_toff, for kernels that copy them)
  // the same two tables packed, one 32-bit entry per term: index | code << 16, code = position of the coefficient in the context's
  // dictionaries (MpxFusedArgs::l_dict / m_dict); fused kernels built with MPX_FUSE_NDICT_LOC / _MU > 0 read these (half the loads)
  const uint32_t* loc_pack;
  const uint32_t* mu_pack;
};

// Per-call arguments of the fused point kernels (all sets of the context in one launch).
struct MpxPtCall {
  const MpxPtSet* sets;
  int32_t n_sets, n_g, B, b_per_block;
  const double* z;
  int64_t z_stride;
  const double* lam;
  int64_t lam_stride;
  const double* sigma;
  double* raw;
  int64_t raw_stride;
};

#define MPX_GATHER_LONG 24  // rows with more terms are summed by a whole wavefront (default; MpxGatherArgs::long_threshold)

struct MpxGatherArgs {
  int64_t n_rows;
  const int64_t* ptr;
  const int32_t* src;
  const double* coef;
  const int64_t* long_rows;  // rows with more than MPX_GATHER_LONG terms
  int64_t n_long;
  int32_t n_short_blocks;    // blocks [0, n_short_blocks): one lane per row; the rest: one wavefront per long row
  const double* raw;
  int64_t raw_stride;
  const double* z;
  int64_t z_stride;
  int64_t seg_begin[5];     // row ranges of up to four output arrays
  double* seg_out[4];       // NULL: not requested
  int64_t seg_stride[4];
  int32_t n_seg, B, b_per_block;
  int32_t long_threshold, pad_;  // rows with more terms than this are the long rows of this pass
};
// Arguments of the fused kernels of assembled contexts (mpx_assembly_fused.h; device arrays are built once per context by
// mpx_assembly.cpp).
struct MpxFusedArgs {
  const MpxPtSet* sets;
  int32_t n_sets, n_blocks;  // 64-point blocks over all sets (MpxPtSet::block_first)
  int32_t n_g, B;
  const double* z;
  int64_t z_stride;
  const double* lam;
  int64_t lam_stride;
  const double* sigma;
  // rows of this pass, all output arrays concatenated (f | g | grad_f | jac_val, or hess_val)
  const int32_t* r_idx;   // position in V of the first term (the 1.0 slot for rows without terms)
  const double* r_coef;   // its coefficient (0 for rows without terms)
  const int32_t* r_nt;    // number of terms
  const int64_t* ptr;     // CSR of all terms, positions in V
  const int32_t* idx;
  const double* coef;
  const int32_t* multi_rows;  // rows with 2 .. MT terms (MT = ELL width of the pass, <= 8): m_idx / m_coef [t][n_multi], padded
  const int32_t* m_idx;
  const double* m_coef;
  const int32_t* mid_rows;    // rows with MT + 1 .. MPX_GATHER_LONG terms
  const int32_t* long_rows;   // rows with more
  int32_t n_multi, n_mid, n_long, pad_;
  double* out[4];
  int64_t out_stride[4];
  long long* dbg;  // MPX_FUSE_DEBUG: phase stamps of one workgroup (wall_clock64, 100 MHz), else NULL
  // point-phase schedule: wavefront w runs the tasks task_list[task_ptr[w] .. task_ptr[w + 1]), task = u * n_blocks + block
  // (longest-processing-time-first assignment by the host: the blocks of a chunk differ 5x in cost)
  const int32_t* task_ptr;
  const int32_t* task_list;
  // shared term lists of the chained local variables: chain c = terms ch_ptr[c] .. ch_ptr[c + 1) (z index, coefficient), its partial
  // sums occupy the LDS chain slots ch_slot[c] .. ch_slot[c] + length (slot 0 of a chain = the empty sum)
  const int32_t* ch_ptr;
  const int32_t* ch_idx;
  const double* ch_coef;
  const int32_t* ch_slot;
  int32_t n_chains, pad2_;
  // single-term rows, packed: position in V | code << 16, code = index of the coefficient in r_dict (the distinct coefficients of the
  // pass's rows with at most one term; 0xffffffff: the row has more terms).  One register per row instead of three in the fused
  // kernels (code objects built with MPX_FUSE_NDICT_* > 0; mpx_fuse_info[5..6])
  const uint32_t* r_pack;
  const double* r_dict;
  int32_t n_dict, pad3_;
  const uint32_t* m_pack;  // [t][n_multi] ELL table of the multi-term rows, packed with the same dictionary
  const double* l_dict;    // dictionaries of MpxPtSet::loc_pack / mu_pack (all sets of the context)
  const double* m_dict;
  int32_t n_ldict, n_mdict;
};

// Arguments of the lane-per-evaluation-point kernels of assembled contexts (mpx_assembly_lanes.h, generated by
// mpopt_amd/assembly_lanes.py): every table of a pass is in the instruction stream / constant data of the code object, the call
// carries the arrays.  z, lam_g and the outputs are dense (row strides n_z, n_g, array lengths: compile-time constants there).
struct MpxLaneArgs {
  const double* z;
  const double* lam;
  const double* sigma;
  double* out[4];   // hess_l pass: {hess_val}; first-order pass: {f, g, grad_f, jac_val}
  double* scratch;  // [block][slot][64]: raw values the global rows read (written by the group kernel, read by the *_global kernel)
  int32_t B, n_blocks;  // evaluation points; 64-point blocks
  int32_t order, pad_;  // workgroup -> (group, block): 0 the groups of a block next to each other, 1 the blocks of a group next to each other
};

#if defined(__HIPCC__)
// Wavefront total of a double in every lane, on the DPP path (inclusive scan by row shifts and row broadcasts, then lane 63): a
// fixed tree without LDS round trips -- the long rows of assembled contexts (one wavefront per row: mpx_gather_kernel and the fused
// kernels use the SAME tree so that both paths stay bit-identical) were reduced by a __shfl_down tree, twelve ds_bpermute per sum.
#ifndef MPX_ASM_SUM_DPP
#define MPX_ASM_SUM_DPP 1
#endif
#define MPX_DPP_STEP_(x, CTRL, ROWS)                                                                         \
  x += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROWS, 0xf, false),         \
                        __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROWS, 0xf, false))
__device__ __forceinline__ double mpx_wave_total(double x) {
#if MPX_ASM_SUM_DPP
  MPX_DPP_STEP_(x, 0x111, 0xf);  // row_shr:1
  MPX_DPP_STEP_(x, 0x112, 0xf);  // row_shr:2
  MPX_DPP_STEP_(x, 0x114, 0xf);  // row_shr:4
  MPX_DPP_STEP_(x, 0x118, 0xf);  // row_shr:8
  MPX_DPP_STEP_(x, 0x142, 0xa);  // row_bcast:15 into rows 1 and 3
  MPX_DPP_STEP_(x, 0x143, 0xc);  // row_bcast:31 into rows 2 and 3
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
#else
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
  return __shfl(x, 0, 64);
#endif
}
#endif

#endif
