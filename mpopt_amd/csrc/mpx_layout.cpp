// mpx_layout.cpp -- libmpx planner: problem structure -> tiles, index maps, COO patterns, light / span plans (host only; split out of
// mpx_host.cpp in round 5).  Called once per context by mpx_create.
//
// Reference behaviour restated here (structure only -- all arithmetic is in mpx_kernels.h):
//   decision vector layout          mpopt.py:537-543, 627   (state-major, phases concatenated)
//   constraint row order            mpopt.py:458, 617-621   ([F;C;DU;mU;dU;TC] per phase, events)
//   composite D / W / interpolation mpopt.py:4015-4131      (never formed densely: per-degree
//                                                            tables + per-node (segment, point))
//   node ownership                  mpopt.py:189-195, 208   (shared node belongs to the earlier
//                                                            segment; later segments drop w_0)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <string>
#include <vector>

#include "mpx.h"
#include "mpx_device.h"
#include "mpx_internal.h"

using namespace mpxi;

namespace {

// global z index of a node variable
inline int64_t zcol(const mpx_ctx& c, const PhaseStruct& P, int kind, int comp, int64_t i) {
  const int64_t N = c.N;
  switch (kind) {
    case MPX_COL_X: return P.z_off + (int64_t)comp * N + i;
    case MPX_COL_U: return P.z_off + (int64_t)(c.nx + comp) * N + i;
    case MPX_COL_T0: return P.z_off + (int64_t)(c.nx + c.nu) * N;
    case MPX_COL_TF: return P.z_off + (int64_t)(c.nx + c.nu) * N + 1;
    default: return P.z_off + (int64_t)(c.nx + c.nu) * N + 2 + comp;
  }
}
inline int64_t zterm(const mpx_ctx& c, const PhaseStruct& P, int kind, int comp) {
  switch (kind) {
    case MPX_TV_XF: return zcol(c, P, MPX_COL_X, comp, c.N - 1);
    case MPX_TV_X0: return zcol(c, P, MPX_COL_X, comp, 0);
    case MPX_TV_TF: return zcol(c, P, MPX_COL_TF, 0, 0);
    case MPX_TV_T0: return zcol(c, P, MPX_COL_T0, 0, 0);
    default: return zcol(c, P, MPX_COL_A, comp, 0);
  }
}

// position of (slot q, lane l) inside a tile block of n lanes and ns slots: slot pairs interleaved
// so that a lane's two values are adjacent (16-byte stores, see scatter_slots in mpx_kernels.h)
inline int64_t slot_index(int64_t q, int64_t l, int64_t n, int64_t ns) {
  if ((ns & 1) && q == ns - 1) return q * n + l;
  return (q >> 1) * 2 * n + 2 * l + (q & 1);
}
}  // namespace

namespace mpxi {

int parse_structure(mpx_ctx* c, const int32_t* s, int64_t len) {
  int64_t q = 0;
  auto need = [&](int64_t n) { return q + n <= len; };
  for (int p = 0; p < c->n_phases; ++p) {
    PhaseStruct& P = c->ph[p];
    if (!need(6)) return fail(c, MPX_ERR_INVALID, "structure truncated (phase %d header)", p);
    P.nc = s[q++];
    P.ntc = s[q++];
    P.diff_u = s[q++];
    P.midu = s[q++];
    P.du_cont = s[q++];
    auto read4 = [&](std::vector<Entry4>& v) -> bool {
      if (!need(1)) return false;
      int n = s[q++];
      if (n < 0 || !need(4LL * n)) return false;
      v.resize(n);
      for (int e = 0; e < n; ++e) {
        v[e] = {s[q], s[q + 1], s[q + 2], s[q + 3]};
        q += 4;
      }
      return true;
    };
    if (!read4(P.jv) || !read4(P.hn) || !read4(P.hc)) return fail(c, MPX_ERR_INVALID, "structure truncated (phase %d)", p);
    if (!need(1)) return fail(c, MPX_ERR_INVALID, "structure truncated");
    int n = s[q++];
    if (n < 0 || !need(2LL * n)) return fail(c, MPX_ERR_INVALID, "structure truncated");
    for (int e = 0; e < n; ++e, q += 2) P.mg.push_back({s[q], s[q + 1]});
    if (!need(1)) return fail(c, MPX_ERR_INVALID, "structure truncated");
    n = s[q++];
    if (n < 0 || !need(3LL * n)) return fail(c, MPX_ERR_INVALID, "structure truncated");
    for (int e = 0; e < n; ++e, q += 3) P.tj.push_back({s[q], s[q + 1], s[q + 2]});
    if (!read4(P.th)) return fail(c, MPX_ERR_INVALID, "structure truncated (phase %d terminal)", p);
    // validate kinds / components
    for (auto& e : P.jv) {
      if ((e.a != MPX_ROW_F && e.a != MPX_ROW_C) || e.b < 0 || e.b >= (e.a == MPX_ROW_F ? c->nx : P.nc) || e.c < 0 ||
          e.c > MPX_COL_A)
        return fail(c, MPX_ERR_INVALID, "bad Jacobian entry in phase %d", p);
    }
  }
  if (q != len) return fail(c, MPX_ERR_INVALID, "structure has %lld trailing words", (long long)(len - q));
  return MPX_OK;
}

int build_tables(mpx_ctx* c) {
  std::vector<int> distinct(c->orders.begin(), c->orders.end());
  std::sort(distinct.begin(), distinct.end());
  distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
  for (int d : distinct) {
    if (d < 1 || d > 255) return fail(c, MPX_ERR_UNSUPPORTED, "polynomial degree %d outside 1..255", d);
    {  // what the node kernel of this degree keeps in LDS (mpx_kernels.h: sD, sC for degrees 13 .. stream_above -- higher degrees
       // stream their tables from global memory --; the double-buffered X/U tile)
      const int64_t P1 = d + 1, segs = MPX_TILE / d;
      const int64_t lds = 8 * ((d > 12 && d <= c->stream_above ? P1 * P1 + (int64_t)d * P1 : 0) + 2 * (int64_t)(c->nx + c->nu) * segs * P1) + 8 * 2 * 4 * 64;
      if (lds > 150 * 1024)
        return fail(c, MPX_ERR_UNSUPPORTED, "polynomial degree %d needs %lld KB of LDS per workgroup (%sthe state/control tile); the limit is 150 of the "
                    "160 KB of an MI355X compute unit", d, (long long)(lds / 1024), d <= c->stream_above ? "differentiation + mid-point tables (MPX_TABLES_STREAM_ABOVE is set above this degree) and " : "");
    }
    DegTable t;
    t.deg = d;
    int n = mpx_colloc_n_nodes(c->scheme, d);
    if (n != d + 1) return fail(c, MPX_ERR_UNSUPPORTED, "scheme %d does not give degree+1 nodes (SURVEY a4)", c->scheme);
    t.roots.resize(n);
    mpx_colloc_roots(c->scheme, d, c->tau0, c->tau1, t.roots.data());
    t.D.resize((size_t)n * n);
    mpx_colloc_diff_matrix(t.roots.data(), n, nullptr, 0, 1, t.D.data());
    t.w.resize(n);
    mpx_colloc_quad_weights(t.roots.data(), n, c->tau0, c->tau1, t.w.data());
    std::vector<double> mids(d);
    for (int k = 0; k < d; ++k) mids[k] = (t.roots[k] + t.roots[k + 1]) / 2.0;  // mpopt.py:350-352
    t.Cmid.resize((size_t)d * n);
    mpx_colloc_interp_matrix(t.roots.data(), n, mids.data(), d, t.Cmid.data());
    t.tk.resize(n);
    for (int k = 0; k < n; ++k) t.tk[k] = (t.roots[k] - c->tau0) / (c->tau1 - c->tau0);
    // derivative rows and normalised positions of the same mid-points (residuals fused into the hess_l pass, MPX_MID_RESID;
    // what mpx_resid_plan_create computes for the target points (tau_{k-1} + tau_k) / 2, mpopt.py:1428-1543)
    t.Dmid.resize((size_t)d * n);
    mpx_colloc_diff_matrix(t.roots.data(), n, mids.data(), d, 1, t.Dmid.data());
    t.tkm.resize(d);
    for (int k = 0; k < d; ++k) t.tkm[k] = (mids[k] - c->tau0) / (c->tau1 - c->tau0);
    c->degs.push_back(std::move(t));
  }
  return MPX_OK;
}

int deg_index(const mpx_ctx* c, int d) {
  for (size_t k = 0; k < c->degs.size(); ++k)
    if (c->degs[k].deg == d) return (int)k;
  return -1;
}

int build_layout(mpx_ctx* c) {
  const int nx = c->nx, nu = c->nu, na = c->na, S = c->S;
  c->seg_start.resize(S + 1);
  c->seg_start[0] = 0;
  for (int s = 0; s < S; ++s) c->seg_start[s + 1] = c->seg_start[s] + c->orders[s];
  const int64_t N = c->N = c->seg_start[S] + 1;
  c->n_zp = N * (nx + nu) + 2 + na;
  c->n_z = c->n_zp * c->n_phases;
  c->n_p = (int64_t)S * c->n_phases;
  // composite quadrature weights (mpopt.py:4060-4062): w0 of segment 0, then w[1:] of every segment
  c->compW.resize(N);
  c->compW[0] = c->degs[deg_index(c, c->orders[0])].w[0];
  for (int s = 0; s < S; ++s) {
    const DegTable& t = c->degs[deg_index(c, c->orders[s])];
    for (int k = 1; k <= c->orders[s]; ++k) c->compW[c->seg_start[s] + k] = t.w[k];
  }
  // g rows
  int64_t g = 0;
  for (int p = 0; p < c->n_phases; ++p) {
    PhaseStruct& P = c->ph[p];
    P.z_off = c->n_zp * p;
    P.g_off_F = g;
    g += (int64_t)nx * N;
    P.g_off_C = g;
    g += (int64_t)P.nc * N;
    P.g_off_DU = g;
    if (P.diff_u) g += (int64_t)nu * N;
    P.g_off_mU = g;
    if (P.midu) g += (int64_t)nu * (N - 1);
    P.g_off_dU = g;
    if (P.du_cont && S > 1) g += (int64_t)nu * (S - 1);
    P.g_off_TC = g;
    g += P.ntc;
  }
  const int64_t g_events = g;
  const int nl = (int)c->links.size() / 2;
  if (c->n_phases > 1) g += (int64_t)nl * (nx + nu + 1);
  c->n_g = g;

  // tiles: per phase, per degree bucket, whole segments, <= MPX_TILE nodes
  c->tiles.clear();
  for (int p = 0; p < c->n_phases; ++p) {
    PhaseStruct& P = c->ph[p];
    P.tile_first = (int)c->tiles.size();
    for (size_t dt = 0; dt < c->degs.size(); ++dt) {
      const int d = c->degs[dt].deg;
      Bucket B;
      B.phase = p;
      B.deg = d;
      B.dt = (int)dt;
      B.tile_first = (int)c->tiles.size();
      // node list of the bucket: node 0 (if segment 0 has this degree), then points 1..d of every segment
      std::vector<int> segs;
      for (int s = 0; s < S; ++s)
        if (c->orders[s] == d) segs.push_back(s);
      if (segs.empty()) continue;
      const bool has0 = segs[0] == 0;
      if (has0) {
        B.node_i.push_back(0);
        B.node_sk.push_back(0);
      }
      for (int s : segs)
        for (int k = 1; k <= d; ++k) {
          B.node_i.push_back(c->seg_start[s] + k);
          B.node_sk.push_back((s << 8) | k);
        }
      if (has0) {  // mini-tile of node 0: lanes 0..d stage segment 0 in LDS, lane 0 owns the outputs
        MpxTile t{};
        t.m0 = 0;
        t.n = d + 1;
        t.n_own = 1;
        t.node0 = 1;
        t.seg0 = 0;
        t.tile_id = (int32_t)c->tiles.size();
        c->tiles.push_back(t);
      }
      // regular tiles: whole segments, an even number of nodes so that 16-byte stores stay aligned
      int per = MPX_TILE / d;
      if ((d & 1) && (per & 1) && per > 1) --per;
      for (size_t q = 0; q < segs.size(); q += per) {
        const int cnt = (int)std::min<size_t>(per, segs.size() - q);
        MpxTile t{};
        t.m0 = (int32_t)((has0 ? 1 : 0) + q * d);
        t.n = t.n_own = cnt * d;
        t.node0 = 0;
        t.seg0 = segs[q];
        t.tile_id = (int32_t)c->tiles.size();
        c->tiles.push_back(t);
      }
      B.tile_count = (int)c->tiles.size() - B.tile_first;
      if (B.tile_count > 0) c->buckets.push_back(std::move(B));
    }
    P.tile_count = (int)c->tiles.size() - P.tile_first;
  }
  c->tile_begin = 0;
  c->tile_end = (int64_t)c->tiles.size();

  // ---- light plan (mpx_light_*): one high degree on the matrix cores, the low-degree segments in between by lanes --------------
  {
    mpx_ctx::LightPlan& L = c->lplan;
    L = mpx_ctx::LightPlan();
    int n_high = 0, dL = 0;
    bool low_ok = c->degs.size() <= MPX_LIGHT_MAXDEG;
    for (auto& t : c->degs) {
      if (t.deg > 12 && t.deg <= 31) ++n_high, dL = t.deg;
      else if (t.deg > 12) low_ok = false;
    }
    if (n_high == 1 && low_ok) {
      std::vector<int> segsL;
      for (int s = 0; s < S; ++s)
        if (c->orders[s] == dL) segsL.push_back(s);
      const int nL = (int)segsL.size(), nin = nx + nu;
      L.fD_off.assign(c->degs.size(), -1), L.fC_off.assign(c->degs.size(), -1);
      for (size_t k = 0; k < c->degs.size(); ++k) {
        if (c->degs[k].deg == dL) continue;
        L.fD_off[k] = (int32_t)L.ftab.size(), L.ftab.insert(L.ftab.end(), c->degs[k].D.begin(), c->degs[k].D.end());
        L.fC_off[k] = (int32_t)L.ftab.size(), L.ftab.insert(L.ftab.end(), c->degs[k].Cmid.begin(), c->degs[k].Cmid.end());
      }
      // LDS of a workgroup: span rows of its wavefronts + the low-degree tables + 9 KB of static tables, under the 64 KB a launch gets
      // by default (two workgroups per compute unit)
      const int cap_limit = std::min(64 * MPX_LIGHT_CHUNKS, (int)((int64_t)(53248 - 8 * (int64_t)L.ftab.size()) / (8 * MPX_LIGHT_WAVES * nin)));  // the span rows of a workgroup's wavefronts in 52 KB of LDS (+ 9 KB of tables: under the 64 KB a launch gets by default, two workgroups per compute unit)
      L.deg = dL, L.dt = deg_index(c, dL), L.first_node = segsL[0] == 0 ? 1 : 0, L.ok = true;
      for (int i = 0; i < nL && L.ok;) {
        int seg_cap = 16;  // high-degree segments of a group = columns of the matrix instructions (MPX_LIGHT_SEGS=n: fewer, A/B)
        if (const char* e = getenv("MPX_LIGHT_SEGS")) seg_cap = std::min(16, std::max(1, atoi(e)));
        int cnt = std::min(seg_cap, nL - i);
        for (; cnt > 0; --cnt) {  // as many segments as the span buffer and the foreign-node slots (2 turns of 64 lanes) hold
          const int64_t lo_w = i == 0 ? 0 : (int64_t)c->seg_start[segsL[i]] + 1, hi = i + cnt < nL ? (int64_t)c->seg_start[segsL[i + cnt]] + 1 : N;
          const int64_t lo_r = std::max<int64_t>(lo_w - 1, 0);
          int64_t nf = 0;
          for (int s = (i == 0 ? 0 : segsL[i]); s < S && c->seg_start[s] + 1 < hi; ++s)
            if (c->orders[s] != dL) nf += c->orders[s] + (s == 0 ? 1 : 0);
          if (hi - lo_r <= cap_limit && nf <= 128) break;
        }
        if (cnt == 0) { L.ok = false; break; }
        MpxLightGroup Gp{};
        const int64_t lo_w = i == 0 ? 0 : (int64_t)c->seg_start[segsL[i]] + 1, hi = i + cnt < nL ? (int64_t)c->seg_start[segsL[i + cnt]] + 1 : N;
        Gp.lo_w = (int32_t)lo_w, Gp.len_w = (int32_t)(hi - lo_w), Gp.lo_r = (int32_t)std::max<int64_t>(lo_w - 1, 0), Gp.len_r = (int32_t)(hi - Gp.lo_r);
        Gp.seg_first = i, Gp.n_light = cnt, Gp.f_first = (int32_t)L.foreign.size();
        for (int s = (i == 0 ? 0 : segsL[i]); s < S && c->seg_start[s] + 1 < hi; ++s) {
          if (c->orders[s] == dL) continue;
          for (int k = (s == 0 ? 0 : 1); k <= c->orders[s]; ++k)
            L.foreign.push_back(MpxLightForeign{c->seg_start[s] + k - Gp.lo_r, c->seg_start[s] - Gp.lo_r, (deg_index(c, c->orders[s]) << 8) | k, s,
                                                c->degs[deg_index(c, c->orders[s])].tk[k], c->compW[c->seg_start[s] + k]});
        }
        Gp.f_count = (int32_t)L.foreign.size() - Gp.f_first;
        L.span_cap = std::max(L.span_cap, (int)Gp.len_r);
        L.groups.push_back(Gp);
        i += cnt;
      }
      L.span_cap += L.span_cap & 1;
      // (a group's sums go to the partial-sum slot tile_first + group of its phase)
      for (int p = 0; p < c->n_phases && L.ok; ++p) L.ok = (int)L.groups.size() <= c->ph[p].tile_count;
    }
    if (!L.ok) L.groups.clear(), L.foreign.clear(), L.ftab.clear();
    // single-degree grids of low degree (mpx_lightlow_*, light_low_body): spans of `own` nodes, the kernel's compile-time geometry
    if (!L.ok && c->degs.size() == 1 && c->degs[0].deg <= 12) {
      const int P = c->degs[0].deg, cap0 = 53248 / (8 * MPX_LIGHT_WAVES * (nx + nu));
      int max_chl = MPX_LOW_MAX_CHUNKS;
      if (const char* e = getenv("MPX_LOW_MAX_CHUNKS")) max_chl = std::max(1, atoi(e));  // (A/B builds: kernels compiled with -DMPX_LOW_MAX_CHUNKS=n)
      const int chl = std::min(max_chl, std::max(1, (cap0 - 2 * P - 8) / 64));
      L.low = true, L.deg = P, L.dt = 0, L.own = 64 * chl, L.span_cap = (L.own + 2 * P + 8 + 1) & ~1;
      L.n_low_groups = (int)((N + L.own - 1) / L.own);
      L.n_low_chunks = (int)((N + 63) / 64);  // partial-sum slots of a phase in a light pass: one per 64-node chunk
      L.ok = chl >= 2;  // (rows of more than ~24 inputs leave one chunk per span: the node kernels do as well)
    }
    // single-degree grids of HIGH degree (mpx_lighthigh_*, light_high_body; round 6): a workgroup = (segment, 16 evaluation points), the
    // input tile [nx + nu][4 ceil((P + 1) / 4)][17] doubles in LDS, one partial-sum slot per segment
    if (!L.ok && c->degs.size() == 1 && c->degs[0].deg >= 32 && !getenv("MPX_NO_LIGHT_HIGH")) {
      const int P = c->degs[0].deg, kp = 4 * ((P + 1 + 3) / 4);
      L = mpx_ctx::LightPlan();
      L.high = true, L.deg = P, L.dt = 0, L.span_cap = kp, L.n_low_chunks = S, L.n_low_groups = S;
      L.ok = (int64_t)(nx + nu) * kp * 17 * 8 + 8192 <= 150 * 1024;
    }
  }

  // ---- packed g / grad_f staging (see MpxIO::gtmp): used by mixed-degree phases and by segment-sharded evaluations ----
  {
    std::vector<int> per_phase(c->n_phases, 0);
    for (auto& B : c->buckets) per_phase[B.phase]++;
    c->g_packed = false;  // mixed-degree phase present: full evaluations stage g / grad_f through the packed block
    for (int v : per_phase) c->g_packed = c->g_packed || v > 1;
    c->gmap.assign((size_t)c->n_g, -1);
    c->qmap.assign((size_t)c->n_z, -1);
    c->tile_g_size.assign(c->tiles.size(), 0);
    std::vector<int64_t> stage_pos((size_t)c->n_phases * N, -1);  // node -> slot-0 position of its staged values
    std::vector<int32_t> stage_n((size_t)c->n_phases * N, 0);     //         slot stride (own lanes of its tile)
    int64_t pos = 0;
    for (auto& B : c->buckets) {
      const PhaseStruct& P = c->ph[B.phase];
      const int sC = nx, sDU = nx + P.nc, sMU = sDU + (P.diff_u ? nu : 0), sQ = sMU + (P.midu ? nu : 0), nsg = sQ + nx + nu;
      for (int t = B.tile_first; t < B.tile_first + B.tile_count; ++t) {
        MpxTile& T = c->tiles[t];
        T.g_base = pos;
        const int64_t n = T.n_own;
        for (int64_t l = 0; l < n; ++l) {
          const int64_t i = B.node_i[T.m0 + l];
          const int k = B.node_sk[T.m0 + l] & 255;
          for (int a = 0; a < nx; ++a) c->gmap[P.g_off_F + (int64_t)a * N + i] = pos + a * n + l;
          for (int j = 0; j < P.nc; ++j) c->gmap[P.g_off_C + (int64_t)j * N + i] = pos + (sC + j) * n + l;
          if (P.diff_u)
            for (int q = 0; q < nu; ++q) c->gmap[P.g_off_DU + (int64_t)q * N + i] = pos + (sDU + q) * n + l;
          if (P.midu && k >= 1)
            for (int q = 0; q < nu; ++q) c->gmap[P.g_off_mU + (int64_t)q * (N - 1) + (i - 1)] = pos + (sMU + q) * n + l;
          for (int a = 0; a < nx + nu; ++a) c->qmap[P.z_off + (int64_t)a * N + i] = pos + (sQ + a) * n + l;
          stage_pos[(size_t)B.phase * N + i] = pos + l, stage_n[(size_t)B.phase * N + i] = (int32_t)n;
        }
        c->tile_g_size[t] = (int64_t)nsg * n;
        pos += (int64_t)nsg * n;
      }
    }
    c->gtmp_n = pos;

    // Absorbing buckets: per phase the bucket with the most nodes.  Its tiles (in segment order) cut [0, N) into contiguous
    // spans -- tile j from its first own node up to the first own node of tile j + 1 -- and write the g / grad_f rows of their
    // span completely: own nodes from registers, the nodes of other buckets in between from the staging block those buckets'
    // kernels (launched earlier) filled.  The unpack pass is then not needed.  Limits: the span rows live in LDS (<= 32 KB per
    // workgroup) and a tile's foreign nodes are fetched by one lane each (<= MPX_TILE); grids outside them keep the unpack pass.
    c->absorb = c->g_packed && !getenv("MPX_NO_ABSORB");
    c->abs_fpos.clear(), c->abs_fstage.clear(), c->abs_fn.clear();
    for (auto& T : c->tiles) T.span_lo = T.span_len = T.f_first = T.f_count = 0;
    for (int p = 0; p < c->n_phases && c->absorb; ++p) {
      const PhaseStruct& P = c->ph[p];
      Bucket* best = nullptr;
      for (auto& B : c->buckets)
        if (B.phase == p && (!best || B.node_i.size() > best->node_i.size())) best = &B;
      if (!best) continue;
      Bucket& B = *best;
      std::vector<int> ts;  // the bucket's tiles of whole segments (not the node-0 mini tile), in segment order already
      for (int t = B.tile_first; t < B.tile_first + B.tile_count; ++t)
        if (!c->tiles[t].node0) ts.push_back(t);
      if (ts.empty()) { c->absorb = false; break; }
      const int nsg = nx + P.nc + (P.diff_u ? nu : 0) + (P.midu ? nu : 0) + nx + nu;
      int cap = 0;
      for (size_t j = 0; j < ts.size(); ++j) {
        MpxTile& T = c->tiles[ts[j]];
        const int64_t lo = j == 0 ? 0 : (int64_t)c->seg_start[T.seg0] + 1;
        const int64_t hi = j + 1 < ts.size() ? (int64_t)c->seg_start[c->tiles[ts[j + 1]].seg0] + 1 : N;
        T.span_lo = (int32_t)lo, T.span_len = (int32_t)(hi - lo), T.f_first = (int32_t)c->abs_fpos.size();
        std::vector<char> mine((size_t)(hi - lo), 0);
        for (int l = 0; l < T.n_own; ++l) mine[(size_t)(B.node_i[T.m0 + l] - lo)] = 1;
        for (int64_t i = lo; i < hi; ++i)
          if (!mine[(size_t)(i - lo)]) {
            c->abs_fpos.push_back((int32_t)(i - lo));
            c->abs_fstage.push_back(stage_pos[(size_t)p * N + i]);
            c->abs_fn.push_back(stage_n[(size_t)p * N + i]);
          }
        T.f_count = (int32_t)c->abs_fpos.size() - T.f_first;
        cap = std::max(cap, (int)T.span_len);
        if (T.f_count > MPX_TILE) {
          if (c->absorb) c->notes += "mixed-degree grid: a tile of the largest bucket would have to fetch " + std::to_string(T.f_count) + " nodes of other buckets (limit " +
                                     std::to_string(MPX_TILE) + "): g / grad_f of the heavy passes go through the staging block and the unpack pass instead of row spans\n";
          c->absorb = false;
        }
      }
      cap += cap & 1;
      {  // static LDS of the bucket's node kernel (same arithmetic as build_tables) + the span rows: up to 150 of the 160 KB of a
         // compute unit (past the 64 KB a launch gets by default load_device raises the kernels' dynamic shared memory limit)
        const int64_t P1 = B.deg + 1, segs = MPX_TILE / B.deg;
        const int64_t lds_static = 8 * ((B.deg > 12 && B.deg <= c->stream_above ? P1 * P1 + (int64_t)B.deg * P1 : 0) + 2 * (int64_t)(nx + nu) * segs * P1) + 8 * 2 * 4 * 64;
        if (lds_static + (int64_t)cap * nsg * 8 > 150 * 1024) {
          if (c->absorb) c->notes += "mixed-degree grid: the row spans of the largest bucket need " + std::to_string((lds_static + (int64_t)cap * nsg * 8) / 1024) +
                                     " KB of LDS per workgroup (limit 150): g / grad_f of the heavy passes go through the staging block and the unpack pass instead\n";
          c->absorb = false;
        }
        B.abs_lds_static = lds_static;
      }
      B.abs_cap = cap, B.abs_slots = nsg;
    }
    if (!c->absorb) {
      for (auto& T : c->tiles) T.span_lo = T.span_len = T.f_first = T.f_count = 0;
      for (auto& B : c->buckets) B.abs_cap = 0;
      c->abs_fpos.clear(), c->abs_fstage.clear(), c->abs_fn.clear();
    }
  }

  // ---- Jacobian pattern -----------------------------------------------------------------
  // value blocks of the tiles: even-sized blocks first so that they all start 16-byte aligned
  std::vector<int32_t>&jr = c->jrow, &jc = c->jcol;
  int64_t jpos = 0;
  auto tile_slots = [&](const MpxTile& T, const PhaseStruct& P, int d) {
    int64_t sl = (int64_t)nx * (d + 1) + (int64_t)P.jv.size() + (P.diff_u ? (int64_t)nu * (d + 1) : 0);
    if (P.midu && !T.node0) sl += (int64_t)nu * (d + 1);
    return sl;
  };
  c->tile_jac_size.assign(c->tiles.size(), 0);
  c->tile_hess_size.assign(c->tiles.size(), 0);
  for (int pass = 0; pass < 2; ++pass)
    for (auto& B : c->buckets)
      for (int t = B.tile_first; t < B.tile_first + B.tile_count; ++t) {
        MpxTile& T = c->tiles[t];
        const int64_t size = tile_slots(T, c->ph[B.phase], B.deg) * T.n_own;
        if ((int)(size & 1) != pass) continue;
        T.jac_base = jpos;
        c->tile_jac_size[t] = size;
        jpos += size;
      }
  jr.assign(jpos, 0);
  jc.assign(jpos, 0);
  // which entries depend on (z, p): the diagonal of a node's D block (D[k][k] - d(h Sx dyn_a)/dX_a; counted as variable whether or
  // not that derivative is structurally zero), the variable entries of the node, the terminal rows -- everything else is a copy of
  // a table entry (node_body: `variable`; MPX_JAC_VARIABLE_ONLY rewrites a superset of this set, never less)
  c->jac_var.assign((size_t)jpos, 0);
  for (auto& B : c->buckets) {
    const PhaseStruct& P = c->ph[B.phase];
    const int d = B.deg, P1 = d + 1;
    for (int t = B.tile_first; t < B.tile_first + B.tile_count; ++t) {
      const MpxTile& T = c->tiles[t];
      const int64_t n = T.n_own, base = T.jac_base;
      for (int64_t l = 0; l < n; ++l) {
        const int64_t i = B.node_i[T.m0 + l];
        const int sk = B.node_sk[T.m0 + l], s = sk >> 8;
        const int64_t st = c->seg_start[s];
        int64_t q = 0;
        const int64_t ns = tile_slots(T, P, d);
        auto put = [&](int64_t row, int64_t col, bool var = false) {
          const int64_t at = base + slot_index(q, l, n, ns);
          jr[at] = (int32_t)row;
          jc[at] = (int32_t)col;
          c->jac_var[(size_t)at] = var ? 1 : 0;
          ++q;
        };
        for (int a = 0; a < nx; ++a)
          for (int j = 0; j < P1; ++j) put(P.g_off_F + (int64_t)a * N + i, zcol(*c, P, MPX_COL_X, a, st + j), j == (sk & 255));
        for (auto& e : P.jv) put((e.a == MPX_ROW_F ? P.g_off_F : P.g_off_C) + (int64_t)e.b * N + i, zcol(*c, P, e.c, e.d, i), true);
        if (P.diff_u)
          for (int u = 0; u < nu; ++u)
            for (int j = 0; j < P1; ++j) put(P.g_off_DU + (int64_t)u * N + i, zcol(*c, P, MPX_COL_U, u, st + j));
        if (P.midu && !T.node0)
          for (int u = 0; u < nu; ++u)
            for (int j = 0; j < P1; ++j) put(P.g_off_mU + (int64_t)u * (N - 1) + (i - 1), zcol(*c, P, MPX_COL_U, u, st + j));
      }
    }
  }
  c->jac_tiles_end = jpos;
  for (int p = 0; p < c->n_phases; ++p) {  // terminal-constraint entries
    PhaseStruct& P = c->ph[p];
    P.jac_TC = jpos;
    for (auto& e : P.tj) {
      jr.push_back((int32_t)(P.g_off_TC + e.row));
      jc.push_back((int32_t)zterm(*c, P, e.kind, e.comp));
      c->jac_var.push_back(1);
      ++jpos;
    }
  }
  // linear rows: control-slope continuity (mpopt.py:398-411), then events (mpopt.py:484-519)
  c->lin_ptr.assign(1, 0);
  c->lin_jac = jpos;
  for (int p = 0; p < c->n_phases; ++p) {
    const PhaseStruct& P = c->ph[p];
    if (!(P.du_cont && S > 1)) continue;
    for (int u = 0; u < nu; ++u)
      for (int s = 0; s + 1 < S; ++s) {
        const DegTable& ta = c->degs[deg_index(c, c->orders[s])];
        const DegTable& tb = c->degs[deg_index(c, c->orders[s + 1])];
        const int pa = ta.deg, pb = tb.deg;
        // end slope of segment s minus start slope of segment s+1; the shared node merges
        for (int j = 0; j <= pa + pb; ++j) {
          double coef = 0;
          if (j <= pa) coef += ta.D[(size_t)pa * (pa + 1) + j];
          if (j >= pa) coef -= tb.D[(size_t)0 * (pb + 1) + (j - pa)];
          c->lin_idx.push_back(zcol(*c, P, MPX_COL_U, u, c->seg_start[s] + j));
          c->lin_coef.push_back(coef);
        }
        c->lin_ptr.push_back((int64_t)c->lin_idx.size());
        c->lin_row.push_back(P.g_off_dU + (int64_t)u * (S - 1) + s);
      }
  }
  if (c->n_phases > 1) {
    int64_t row = g_events;
    for (int blk = 0; blk < 3; ++blk)
      for (int l = 0; l < nl; ++l) {
        const PhaseStruct& Pi = c->ph[c->links[2 * l]];
        const PhaseStruct& Pj = c->ph[c->links[2 * l + 1]];
        int cnt = blk == 0 ? nx : (blk == 1 ? nu : 1);
        for (int a = 0; a < cnt; ++a) {
          if (blk == 2) {  // t0_j - tf_i
            c->lin_idx.push_back(zcol(*c, Pj, MPX_COL_T0, 0, 0));
            c->lin_idx.push_back(zcol(*c, Pi, MPX_COL_TF, 0, 0));
          } else {
            int kind = blk == 0 ? MPX_COL_X : MPX_COL_U;
            c->lin_idx.push_back(zcol(*c, Pj, kind, a, 0));
            c->lin_idx.push_back(zcol(*c, Pi, kind, a, N - 1));
          }
          c->lin_coef.push_back(1.0);
          c->lin_coef.push_back(-1.0);
          c->lin_ptr.push_back((int64_t)c->lin_idx.size());
          c->lin_row.push_back(row++);
        }
      }
  }
  for (size_t r = 0; r + 1 < c->lin_ptr.size(); ++r)
    for (int64_t e = c->lin_ptr[r]; e < c->lin_ptr[r + 1]; ++e) {
      jr.push_back((int32_t)c->lin_row[r]);
      jc.push_back((int32_t)c->lin_idx[e]);
      c->jac_var.push_back(0);
      ++jpos;
    }
  c->nnz_j = jpos;
  {  // the same rows by column (nlp_grad: grad_gamma_x += J^T lam_g of these rows; entries in row order inside a column)
    std::map<int64_t, std::vector<std::pair<int64_t, double>>> cols;
    for (size_t r = 0; r + 1 < c->lin_ptr.size(); ++r)
      for (int64_t e = c->lin_ptr[r]; e < c->lin_ptr[r + 1]; ++e) cols[c->lin_idx[e]].push_back({c->lin_row[r], c->lin_coef[e]});
    c->lt_ptr.assign(1, 0), c->lt_col.clear(), c->lt_row.clear(), c->lt_coef.clear();
    for (auto& kv : cols) {
      c->lt_col.push_back(kv.first);
      for (auto& e : kv.second) c->lt_row.push_back(e.first), c->lt_coef.push_back(e.second);
      c->lt_ptr.push_back((int64_t)c->lt_row.size());
    }
  }

  // ---- node-ordered tiles of the hess_l pass on mixed-degree grids (MpxHTile) ------------------
  c->hess_by_node = c->degs.size() > 1 && !getenv("MPX_NO_HESS_BY_NODE");
  c->htiles.clear(), c->ph_htile_first.assign(c->n_phases, 0), c->ph_htile_count.assign(c->n_phases, 0);
  if (c->hess_by_node) {
    c->node_seg.assign((size_t)N, 0), c->node_tk.assign((size_t)N, 0.0);
    for (int s = 0; s < S; ++s) {
      const DegTable& t = c->degs[deg_index(c, c->orders[s])];
      for (int k = (s == 0 ? 0 : 1); k <= c->orders[s]; ++k) c->node_seg[c->seg_start[s] + k] = s, c->node_tk[c->seg_start[s] + k] = t.tk[k];
    }
    for (int p = 0; p < c->n_phases && c->hess_by_node; ++p) {
      c->ph_htile_first[p] = (int32_t)c->htiles.size();
      for (int64_t i0 = 0; i0 < N; i0 += MPX_TILE) {
        MpxHTile T{};
        T.i0 = (int32_t)i0, T.n = (int32_t)std::min<int64_t>(MPX_TILE, N - i0);
        T.tile_id = c->ph[p].tile_first + (int32_t)(c->htiles.size() - c->ph_htile_first[p]);
        c->htiles.push_back(T);
      }
      c->ph_htile_count[p] = (int32_t)c->htiles.size() - c->ph_htile_first[p];
      if (c->ph_htile_count[p] > c->ph[p].tile_count) c->hess_by_node = false;  // (their partial sums use the phase's tile slots)
    }
    if (!c->hess_by_node) c->htiles.clear();
  }

  // ---- Hessian pattern (upper triangle) -----------------------------------------------------
  std::vector<int32_t>&hr = c->hrow, &hc = c->hcol;
  int64_t hpos = 0;
  std::vector<std::map<std::pair<int64_t, int64_t>, int64_t>> edge(c->n_phases);
  if (c->hess_by_node) {
    for (int pass = 0; pass < 2; ++pass)  // even-sized blocks first: they all start 16-byte aligned
      for (auto& T : c->htiles) {
        int p = 0;
        while (p + 1 < c->n_phases && T.tile_id >= c->ph[p + 1].tile_first) ++p;
        const int64_t size = (int64_t)c->ph[p].hn.size() * T.n;
        if ((int)(size & 1) != pass) continue;
        T.hess_base = hpos;
        hpos += size;
      }
    hr.assign(hpos, 0), hc.assign(hpos, 0);
    for (auto& T : c->htiles) {
      int p = 0;
      while (p + 1 < c->n_phases && T.tile_id >= c->ph[p + 1].tile_first) ++p;
      const PhaseStruct& P = c->ph[p];
      const int64_t ns = (int64_t)P.hn.size();
      for (int64_t l = 0; l < T.n; ++l) {
        const int64_t i = T.i0 + l;
        int64_t q = 0;
        for (auto& e : P.hn) {
          const int64_t r = zcol(*c, P, e.a, e.b, i), cc = zcol(*c, P, e.c, e.d, i);
          const int64_t at = T.hess_base + slot_index(q, l, T.n, ns);
          hr[at] = (int32_t)r, hc[at] = (int32_t)cc;
          if (i == 0 || i == N - 1) edge[p][{r, cc}] = at;
          ++q;
        }
      }
    }
  } else {
  for (int pass = 0; pass < 2; ++pass)
    for (auto& B : c->buckets)
      for (int t = B.tile_first; t < B.tile_first + B.tile_count; ++t) {
        MpxTile& T = c->tiles[t];
        const int64_t size = (int64_t)c->ph[B.phase].hn.size() * T.n_own;
        if ((int)(size & 1) != pass) continue;
        T.hess_base = hpos;
        c->tile_hess_size[t] = size;
        hpos += size;
      }
  hr.assign(hpos, 0);
  hc.assign(hpos, 0);
  for (auto& B : c->buckets) {
    const PhaseStruct& P = c->ph[B.phase];
    for (int t = B.tile_first; t < B.tile_first + B.tile_count; ++t) {
      const MpxTile& T = c->tiles[t];
      const int64_t n = T.n_own, base = T.hess_base;
      for (int64_t l = 0; l < n; ++l) {
        const int64_t i = B.node_i[T.m0 + l];
        int64_t q = 0;
        const int64_t ns = (int64_t)P.hn.size();
        for (auto& e : P.hn) {
          int64_t r = zcol(*c, P, e.a, e.b, i), cc = zcol(*c, P, e.c, e.d, i);
          const int64_t at = base + slot_index(q, l, n, ns);
          hr[at] = (int32_t)r;
          hc[at] = (int32_t)cc;
          if (i == 0 || i == N - 1) edge[B.phase][{r, cc}] = at;
          ++q;
        }
      }
    }
  }
  }  // (bucket-ordered hess_l tiles)
  c->mg_off.assign(MPX_MAX_PHASES, 0);
  c->hc_off.assign(MPX_MAX_PHASES, 0);
  c->th_off.assign(MPX_MAX_PHASES, 0);
  int nred = 1;
  for (int p = 0; p < c->n_phases; ++p) {
    const PhaseStruct& P = c->ph[p];
    nred = std::max(nred, std::max(3 + na, (int)P.hc.size()));
    c->mg_off[p] = (int32_t)c->mg_dst.size();
    for (auto& e : P.mg) c->mg_dst.push_back(zterm(*c, P, e.first, e.second));
    c->hc_off[p] = (int32_t)c->hc_dst.size();
    for (auto& e : P.hc) {
      int64_t r = zcol(*c, P, e.a, e.b, 0), cc = zcol(*c, P, e.c, e.d, 0);
      hr.push_back((int32_t)r);
      hc.push_back((int32_t)cc);
      edge[p][{r, cc}] = hpos;
      c->hc_dst.push_back(hpos++);
    }
    c->th_off[p] = (int32_t)c->th_dst.size();
    for (auto& e : P.th) {
      int64_t r = zterm(*c, P, e.a, e.b), cc = zterm(*c, P, e.c, e.d);
      if (r > cc) std::swap(r, cc);
      auto it = edge[p].find({r, cc});
      if (it != edge[p].end()) {
        c->th_dst.push_back(it->second | MPX_ACCUM_BIT);
      } else {
        hr.push_back((int32_t)r);
        hc.push_back((int32_t)cc);
        edge[p][{r, cc}] = hpos;
        c->th_dst.push_back(hpos++);
      }
    }
  }
  c->nred = nred;
  c->nnz_h = hpos;
  return MPX_OK;
}


}  // namespace mpxi
