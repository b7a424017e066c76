// mpx_shard.cpp -- libmpx: segment sharding of one evaluation over the ranks of a multi-GPU run (include/mpx.h,
// mpx_shard_*).  Split out of mpx_host.cpp in round 5 (the host runtime was one 3000-line translation unit); shares the context
// definition and the helpers of mpx_internal.h.  What shards and why: a node reads only its own segment (mpopt.py:189-198, 227-232),
// phases are coupled by the event rows only (mpopt.py:464-521).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mpx.h"
#include "mpx_device.h"

#include "mpx_internal.h"

using namespace mpxi;

// ---- segment sharding -------------------------------------------------------------------------------
// Ranks evaluate disjoint contiguous tile ranges; what a rank owns afterwards is a handful of contiguous runs: the value
// blocks of its tiles (jac_val or hess_val), its run of the packed g / grad_f staging block and its per-tile partial sums.
// mpx_shard_pack copies the runs into one exchange buffer, the caller all-gathers the buffers (RCCL over xGMI),
// mpx_shard_unpack scatters the other ranks' runs into place, and the MPX_BOUNDARY_ONLY pass finishes the evaluation.
namespace {

// (partials_only: the owner-resident exchange -- a rank's buffer holds nothing but its tile partials, at offset 0)
__global__ __launch_bounds__(256) void mpx_shard_copy_kernel(const MpxShardEnt* __restrict__ ents, int64_t B, int64_t rank_len, int my_rank,
                                                             int unpack, int partials_only, double* vals, double* gtmp, double* partial, double* buf) {
  const MpxShardEnt E = ents[blockIdx.y];
  if (unpack ? E.rank == my_rank : E.rank != my_rank) return;
  if (partials_only && E.kind != 2) return;
  double* __restrict__ base = E.kind == 0 ? vals : (E.kind == 1 ? gtmp : partial);
  if (!base) return;
  double* __restrict__ pk = buf + (unpack ? (int64_t)E.rank * rank_len * B : 0) + (partials_only ? E.part_off : E.dst_off) * B;
  const int64_t n = B * E.len;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / E.len, i = e - b * E.len;
    double* __restrict__ at = base + E.src_off + b * E.stride + i;
    if (unpack)
      *at = pk[e];
    else
      pk[e] = *at;
  }
}

// greedy prefix split of the tiles by weight (Jacobian block size): contiguous ranges, possibly empty
std::vector<int64_t> shard_cuts(const std::vector<int64_t>& w, int world) {
  const int64_t n = (int64_t)w.size();
  std::vector<double> cum(n + 1, 0.0);
  for (int64_t t = 0; t < n; ++t) cum[t + 1] = cum[t] + (double)std::max<int64_t>(w[t], 1);
  std::vector<int64_t> cuts(1, 0);
  for (int r = 1; r < world; ++r) {
    const double target = cum[n] * r / world;
    int64_t k = std::lower_bound(cum.begin(), cum.end(), target) - cum.begin();
    if (k > 0 && std::abs(cum[k - 1] - target) <= std::abs(cum[std::min(k, n)] - target)) --k;
    cuts.push_back(std::max(cuts.back(), std::min(k, n)));
  }
  cuts.push_back(n);
  return cuts;
}

// value runs of the node-ordered hess_l tiles rank r owns (merged where adjacent)
void hess_node_runs(const mpx_ctx* c, int r, std::vector<std::pair<int64_t, int64_t>>& runs) {
  runs.clear();
  for (int p = 0; p < c->n_phases; ++p) {
    const int64_t cnt = c->ph_htile_count[p], a0 = c->shard_cuts_h[r] * cnt / c->shard_cuts_h.back(), a1 = c->shard_cuts_h[r + 1] * cnt / c->shard_cuts_h.back();
    for (int64_t h = c->ph_htile_first[p] + a0; h < c->ph_htile_first[p] + a1; ++h)
      runs.push_back({c->htiles[(size_t)h].hess_base, (int64_t)c->ph[p].hn.size() * c->htiles[(size_t)h].n});
  }
  std::sort(runs.begin(), runs.end());
  size_t o = 0;
  for (size_t k = 0; k < runs.size(); ++k) {
    if (runs[k].second <= 0) continue;
    if (o > 0 && runs[o - 1].first + runs[o - 1].second == runs[k].first)
      runs[o - 1].second += runs[k].second;
    else
      runs[o++] = runs[k];
  }
  runs.resize(o);
}

void shard_runs(const std::vector<MpxTile>& tiles, const std::vector<int64_t>& size, bool hess, int64_t tb, int64_t te,
                std::vector<std::pair<int64_t, int64_t>>& runs) {
  runs.clear();
  for (int64_t t = tb; t < te; ++t)
    if (size[t] > 0) runs.push_back({hess ? tiles[t].hess_base : tiles[t].jac_base, size[t]});
  std::sort(runs.begin(), runs.end());
  size_t o = 0;
  for (size_t k = 0; k < runs.size(); ++k) {
    if (o > 0 && runs[o - 1].first + runs[o - 1].second == runs[k].first)
      runs[o - 1].second += runs[k].second;
    else
      runs[o++] = runs[k];
  }
  runs.resize(o);
}

}  // namespace

extern "C" int mpx_shard_setup(mpx_ctx* c, int world, int rank) {
  if (!c || world < 1 || rank < 0 || rank >= world) return fail(c, MPX_ERR_INVALID, "mpx_shard_setup: bad world / rank");
  if (c->kind != 0) return fail(c, MPX_ERR_UNSUPPORTED, "assembled contexts have no tiles to shard");
  const int64_t nt = (int64_t)c->tiles.size();
  c->shard_world = world;
  c->shard_rank = rank;
  for (int ps = 0; ps < 2; ++ps) {
    c->shard_ent[ps].clear();
    c->shard_ent_first[ps].assign(1, 0);
    c->shard_len[ps] = 0;
    if (c->d_shard_ent[ps]) (void)dev_free(c->d_shard_ent[ps]);
    c->d_shard_ent[ps] = nullptr;
  }
  if (world == 1) {
    c->shard_cuts = {0, nt};
    c->tile_begin = 0, c->tile_end = nt, c->run_boundary = 1;
    return MPX_OK;
  }
  c->shard_cuts = shard_cuts(c->tile_jac_size, world);
  if (c->hess_by_node) {  // hess_l pass: ranks split the node-ordered tiles of every phase in the same proportions
    const int64_t cnt = c->ph_htile_count.empty() ? 0 : c->ph_htile_count[0];
    c->shard_cuts_h.assign(1, 0);
    for (int r = 1; r <= world; ++r) c->shard_cuts_h.push_back(cnt * r / world);
  }
  c->tile_begin = c->shard_cuts[rank];
  c->tile_end = c->shard_cuts[rank + 1];
  c->run_boundary = 0;
  std::vector<std::pair<int64_t, int64_t>> runs;
  c->shard_len_part = 0;  // owner-resident exchange: the tile partials only (max over ranks and passes)
  for (int ps = 0; ps < 2; ++ps) {
    for (int r = 0; r < world; ++r) {
      const int64_t tb = c->shard_cuts[r], te = c->shard_cuts[r + 1];
      int64_t pos = 0, ppos = 0;
      auto add = [&](int kind, int64_t off, int64_t len, int64_t stride) {
        if (len <= 0) return;
        c->shard_ent[ps].push_back(MpxShardEnt{off, len, stride, pos, kind, r, kind == 2 ? ppos : 0});
        pos += len;
        if (kind == 2) ppos += len, c->shard_len_part = std::max(c->shard_len_part, ppos);
      };
      if (ps == 1 && c->hess_by_node) {
        hess_node_runs(c, r, runs);
        for (auto& q : runs) add(0, q.first, q.second, c->nnz_h);
        for (int p = 0; p < c->n_phases; ++p) {  // the partial sums of its tiles: one run of tile slots per phase
          const int64_t cnt = c->ph_htile_count[p], a0 = c->shard_cuts_h[r] * cnt / c->shard_cuts_h.back(), a1 = c->shard_cuts_h[r + 1] * cnt / c->shard_cuts_h.back();
          add(2, (c->ph[p].tile_first + a0) * c->nred, (a1 - a0) * c->nred, nt * c->nred);
        }
        c->shard_len[ps] = std::max(c->shard_len[ps], pos);
        c->shard_ent_first[ps].push_back((int32_t)c->shard_ent[ps].size());
        continue;
      }
      shard_runs(c->tiles, ps ? c->tile_hess_size : c->tile_jac_size, ps == 1, tb, te, runs);
      for (auto& q : runs) add(0, q.first, q.second, ps ? c->nnz_h : c->nnz_j);
      if (ps == 0 && te > tb) add(1, c->tiles[tb].g_base, c->tiles[te - 1].g_base + c->tile_g_size[te - 1] - c->tiles[tb].g_base, c->gtmp_n);
      add(2, tb * c->nred, (te - tb) * c->nred, nt * c->nred);
      c->shard_len[ps] = std::max(c->shard_len[ps], pos);
      c->shard_ent_first[ps].push_back((int32_t)c->shard_ent[ps].size());
    }
    c->shard_len[ps] += c->shard_len[ps] & 1;  // keep every rank's slot 16-byte aligned
    c->shard_len_part += c->shard_len_part & 1;

    if (c->has_device) {
      HIPCHK(c, hipSetDevice(c->device));
      int rc = upload(c, &c->d_shard_ent[ps], c->shard_ent[ps]);
      if (rc) return rc;
    }
  }
  return MPX_OK;
}

extern "C" int mpx_shard_info(const mpx_ctx* c, int mask, int64_t* rank_len, int64_t* n_entries, int64_t* tile_cuts) {
  if (!c || c->shard_cuts.empty()) return MPX_ERR_INVALID;
  const int ps = (mask & MPX_HESS) ? 1 : 0;
  if (rank_len) *rank_len = (mask & MPX_OWNER_RESIDENT) ? c->shard_len_part : c->shard_len[ps];
  if (n_entries) *n_entries = (int64_t)c->shard_ent[ps].size();
  if (tile_cuts) memcpy(tile_cuts, c->shard_cuts.data(), c->shard_cuts.size() * sizeof(int64_t));
  return MPX_OK;
}

extern "C" int mpx_shard_table(const mpx_ctx* c, int mask, int64_t* out) {
  if (!c || !out || c->shard_cuts.empty()) return MPX_ERR_INVALID;
  const int ps = (mask & MPX_HESS) ? 1 : 0;
  for (auto& e : c->shard_ent[ps]) {
    // (with MPX_OWNER_RESIDENT in the mask the last column of a partial-sum run is its offset in the partials-only exchange buffer)
    *out++ = e.rank, *out++ = e.kind, *out++ = e.src_off, *out++ = e.len, *out++ = e.stride, *out++ = ((mask & MPX_OWNER_RESIDENT) && e.kind == 2) ? e.part_off : e.dst_off;
  }
  return MPX_OK;
}

// Runs of `which` (MPX_G / MPX_GRAD / MPX_JAC / MPX_HESS) owned by `rank` in the owner-resident protocol.
extern "C" int mpx_shard_owned(const mpx_ctx* c, int which, int rank, int64_t* n_runs, int64_t* runs) {
  if (!c || !n_runs || c->shard_cuts.empty() || rank < 0 || rank + 1 >= (int)c->shard_cuts.size()) return MPX_ERR_INVALID;
  if (c->kind != 0) return MPX_ERR_UNSUPPORTED;
  const int64_t tb = c->shard_cuts[rank], te = c->shard_cuts[rank + 1];
  std::vector<std::pair<int64_t, int64_t>> out;
  if (which == MPX_HESS && c->hess_by_node) {
    hess_node_runs(c, rank, out);
  } else if (which == MPX_JAC || which == MPX_HESS) {
    shard_runs(c->tiles, which == MPX_HESS ? c->tile_hess_size : c->tile_jac_size, which == MPX_HESS, tb, te, out);
  } else if (which == MPX_G || which == MPX_GRAD) {
    // the packed staging map of build_layout knows which tile holds every node row: rows whose staged position falls into the
    // staging run of the rank's tiles belong to the rank
    std::vector<int64_t> rows;
    if (te > tb) {
      const int64_t lo = c->tiles[tb].g_base, hi = c->tiles[te - 1].g_base + c->tile_g_size[te - 1];
      const std::vector<int64_t>& map = which == MPX_G ? c->gmap : c->qmap;
      for (int64_t r = 0; r < (int64_t)map.size(); ++r)
        if (map[r] >= lo && map[r] < hi) rows.push_back(r);
    }
    for (int64_t r : rows) {
      if (!out.empty() && out.back().first + out.back().second == r)
        ++out.back().second;
      else
        out.push_back({r, 1});
    }
  } else {
    return MPX_ERR_INVALID;
  }
  *n_runs = (int64_t)out.size();
  if (runs)
    for (auto& q : out) *runs++ = q.first, *runs++ = q.second;
  return MPX_OK;
}

extern "C" int mpx_device_pci_bus_id(int device, char* out, int len) {
  if (!out || len < 16) return MPX_ERR_INVALID;
  out[0] = 0;
  return hipDeviceGetPCIBusId(out, len, device) == hipSuccess ? MPX_OK : MPX_ERR_HIP;
}

static int shard_copy(mpx_ctx* c, int mask, int64_t batch, double* vals, double* buf, int unpack) {
  if (!c || !buf || batch < 1) return MPX_ERR_INVALID;
  if (!c->has_device) return fail(c, MPX_ERR_NO_DEVICE, "mpx_shard_pack/unpack: context has no device code; there is no CPU fallback");
  if (c->shard_world <= 1) return fail(c, MPX_ERR_INVALID, "mpx_shard_pack/unpack without mpx_shard_setup(world > 1)");
  const int ps = (mask & MPX_HESS) ? 1 : 0;
  const int part_only = (mask & MPX_OWNER_RESIDENT) ? 1 : 0;
  HIPCHK(c, hipSetDevice(c->device));
  int rc;
  if ((rc = reserve(c, c->partial, (size_t)(batch * partial_slots(c) * c->nred)))) return rc;
  double* gt = nullptr;
  if (!part_only && ps == 0 && (mask & (MPX_G | MPX_GRAD))) {
    if ((rc = reserve(c, c->gtmp, (size_t)(batch * c->gtmp_n)))) return rc;
    gt = c->gtmp.p;
  }
  const int32_t first = unpack ? 0 : c->shard_ent_first[ps][c->shard_rank];
  const int32_t count = unpack ? (int32_t)c->shard_ent[ps].size() : c->shard_ent_first[ps][c->shard_rank + 1] - first;
  if (count <= 0) return MPX_OK;
  int64_t longest = 1;
  for (int32_t k = first; k < first + count; ++k)
    if (!part_only || c->shard_ent[ps][k].kind == 2) longest = std::max(longest, c->shard_ent[ps][k].len * batch);
  const unsigned gx = (unsigned)std::min<int64_t>((longest + 255) / 256, 2048);
  hipLaunchKernelGGL(mpx_shard_copy_kernel, dim3(gx, (unsigned)count), dim3(256), 0, c->stream, c->d_shard_ent[ps] + first, batch,
                     part_only ? c->shard_len_part : c->shard_len[ps], c->shard_rank, unpack, part_only,
                     ((mask & (MPX_JAC | MPX_HESS)) && !part_only ? vals : nullptr), gt, c->partial.p, buf);
  HIPCHK(c, hipGetLastError());
  return MPX_OK;
}

extern "C" int mpx_shard_pack(mpx_ctx* c, int mask, int64_t batch, const double* vals, double* send) {
  return shard_copy(c, mask, batch, const_cast<double*>(vals), send, 0);
}

extern "C" int mpx_shard_unpack(mpx_ctx* c, int mask, int64_t batch, const double* recv, double* vals) {
  return shard_copy(c, mask, batch, vals, const_cast<double*>(recv), 1);
}


