// te_shim.hip -- C-ABI of libtravgpu.so (declared in include/travgpu.h).
//
// Owns the device-resident elevation/output layers of a batch of maps and launches the HIP chain.
// No CPU fallback of any kind: without a gfx950 device te_create() fails with TE_ERR_NO_DEVICE.
#include "te_ctx.h"
#include "te_hole_routing.h"

using namespace te;
using namespace te::shim;

namespace te {
namespace shim {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// Build the row-run table of the disc {di^2+dj^2 <= (radius/res)^2}.  Offsets whose squared norm
// equals (radius/res)^2 to within 1e-9 relative are "ties": the reference decides them per cell
// from rounded double positions, so the kernels test them with the same formula.
int build_disc(double radius, double res, Disc* d, const char* what) {
  memset(d, 0, sizeof(*d));
  d->r2 = radius * radius;
  const double q = (radius / res) * (radius / res);
  const double tol = 1e-9 * (q > 1.0 ? q : 1.0);
  const double rmax = sqrt(q + tol);
  if (!(rmax < (double)kMaxRadiusCells + 0.5))
    return fail(TE_ERR_UNSUPPORTED, "%s radius %.6g m is %.2f cells; this build supports up to %d", what, radius,
                radius / res, kMaxRadiusCells);
  const int lim = (int)floor(rmax) + 1;
  d->R = -1;
  d->reach = 0;
  d->npoints = 0;
  for (int b = 0; b <= kMaxRadiusCells; ++b) d->hw[b] = -1;
  for (int b = 0; b <= lim && b <= kMaxRadiusCells; ++b) {
    int hw = -1;
    for (int a = 0; a <= lim; ++a) {
      const double m = (double)(a * a + b * b);
      if (fabs(m - q) <= tol) {
        // tie: all sign combinations, each listed once
        for (int sa = -1; sa <= 1; sa += 2)
          for (int sb = -1; sb <= 1; sb += 2) {
            if ((a == 0 && sa < 0) || (b == 0 && sb < 0)) continue;
            if (d->n_ties >= kMaxTies) return fail(TE_ERR_UNSUPPORTED, "%s radius: too many tie offsets", what);
            d->tie_di[d->n_ties] = (int8_t)(sa * a);
            d->tie_dj[d->n_ties] = (int8_t)(sb * b);
            d->n_ties++;
            const int mx = a > b ? a : b;
            if (mx > d->reach) d->reach = mx;
          }
      } else if (m < q) {
        hw = a;
      }
    }
    d->hw[b] = hw;
    if (hw >= 0) {
      d->R = b;
      d->npoints += (b == 0 ? 1 : 2) * (2 * hw + 1);
    }
  }
  // rows are nested (hw non-increasing) and a tie at (a,b) always sits right after the run end, so
  // runs never skip an interior cell.
  if (d->R > d->reach) d->reach = d->R;
  if (d->hw[0] > d->reach) d->reach = d->hw[0];
  // the shape is named by the largest sum of two squares not above q (only meaningful without ties)
  d->Q = -1;
  if (d->n_ties == 0) {
    int best = -1;
    for (int a = 0; a <= lim; ++a)
      for (int b = 0; b <= lim; ++b) {
        const int m = a * a + b * b;
        if ((double)m < q && m > best) best = m;
      }
    d->Q = best;  // -1: radius below zero cells cannot happen (m = 0 < q unless q == 0, which is a tie)
  }
  return TE_OK;
}

bool same_disc(const Disc& a, const Disc& b) {
  if (a.R != b.R || a.n_ties != b.n_ties) return false;
  for (int k = 0; k <= kMaxRadiusCells; ++k)
    if (a.hw[k] != b.hw[k]) return false;
  if (a.n_ties && a.r2 != b.r2) return false;
  return true;
}

}  // namespace shim
}  // namespace te


// Invalid (non-finite) cells of a layer: one pass at upload time.  out[0]: their number; out[1]: the number of RUNS of them
// in memory order (an invalid cell whose predecessor is valid, or that is the layer's first).  The two pick the march
// k_normals3 uses for strips with invalid cells and its strip height: scattered cells (runs of one) against unobserved
// regions (runs as long as the regions are wide), te_normals3.hip.
__global__ void k_count_invalid(const float* __restrict__ v, size_t n, unsigned long long* __restrict__ out) {
  unsigned cnt = 0, runs = 0;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
    const bool bad = !__builtin_isfinite(v[k]);
    cnt += bad ? 1u : 0u;
    runs += (bad && (k == 0 || __builtin_isfinite(v[k - 1]))) ? 1u : 0u;
  }
  for (int d = 32; d >= 1; d >>= 1) {
    cnt += __shfl_xor((int)cnt, d);
    runs += __shfl_xor((int)runs, d);
  }
  if ((threadIdx.x & 63) == 0 && cnt) {
    atomicAdd(out, (unsigned long long)cnt);
    if (runs) atomicAdd(out + 1, (unsigned long long)runs);
  }
}


namespace te {
namespace shim {
// joins a running prefetch (caller holds c->mu); its result stays in c->prefetch_rc until te_wait_prefetch reports it
void finish_prefetch_locked(te_ctx* c) {
  {
    std::unique_lock<std::mutex> pl(c->pf_mu);
    if (!c->prefetch_running && !c->prefetch_mask) return;
    c->pf_cv.wait(pl, [c] { return !c->prefetch_running; });
  }
  const unsigned mask = c->prefetch_mask;
  c->prefetch_mask = 0;
  c->prefetch_elev = false;
  const bool ok = c->prefetch_rc.load() == TE_OK;
  if (mask & (1u << TE_LAYER_ELEVATION)) {
    // whatever was computed from the previous elevation no longer describes the layer, arrived or torn
    c->chain_done = false;
    c->footprint_done = false;
    c->invalid_cells = -1;
    c->invalid_runs = -1;
    if (ok) {
      c->have_elev = true;
      // the invalid cells are counted like te_upload_elevation counts them (the count picks the normals kernel's march and
      // strip height); a failure leaves the count unknown, which every kernel serves
      if (hipSetDevice(c->device) != hipSuccess || count_invalid_elevation(c) != TE_OK) {
        (void)hipGetLastError();
        c->invalid_cells = -1;
      }
    } else {
      c->have_elev = false;  // partly overwritten: the next chain needs a complete upload
    }
  }
  if (ok) {  // (what an arrived layer changes for the later calls: only once it HAS arrived)
    if (mask & (1u << TE_LAYER_ROBOT_SLOPE)) c->have_robot_slope = true;
    if (mask & (1u << TE_LAYER_TRAVERSABILITY)) c->trav_external = c->trav_ptr_out = true;
  }
}
// layers TE_FILTER_* reads and writes (travgpu.h: TE_FILTER_* table)
unsigned filter_layers(int filter) {
  const unsigned normals = bit(TE_LAYER_NORMAL_X) | bit(TE_LAYER_NORMAL_Y) | bit(TE_LAYER_NORMAL_Z);
  switch (filter) {
    case TE_FILTER_SLOPE: return bit(TE_LAYER_NORMAL_Z) | bit(TE_LAYER_SLOPE);
    case TE_FILTER_STEP: return bit(TE_LAYER_ELEVATION) | bit(TE_LAYER_STEP);
    case TE_FILTER_ROUGHNESS: return bit(TE_LAYER_ELEVATION) | normals | bit(TE_LAYER_ROUGHNESS);
    case TE_FILTER_COMBINE: return bit(TE_LAYER_SLOPE) | bit(TE_LAYER_STEP) | bit(TE_LAYER_ROUGHNESS) | bit(TE_LAYER_TRAVERSABILITY);
    case TE_FILTER_NORMALS: return bit(TE_LAYER_ELEVATION) | normals | bit(TE_LAYER_SLOPE) | bit(TE_LAYER_ROUGHNESS);
    default: return ~0u;
  }
}
}  // namespace shim
}  // namespace te

namespace te {
namespace shim {

// counts the invalid cells of the whole elevation layer on the context's stream and waits for the result
int count_invalid_elevation(te_ctx* c) {
  c->invalid_cells = -1;
  if (!c->d_count) HIP_TRY(hipMalloc((void**)&c->d_count, 2 * sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(c->d_count, 0, 2 * sizeof(unsigned long long), c->stream));
  const size_t n = c->layer_elems;
  int blocks = (int)((n + 256 * 16 - 1) / (256 * 16));
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(k_count_invalid, dim3((unsigned)blocks), dim3(256), 0, c->stream, c->L.elev, n, c->d_count);
  unsigned long long h[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(h, c->d_count, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->invalid_cells = (long long)h[0];
  c->invalid_runs = (long long)h[1];
  return TE_OK;
}

// few invalid cells, scattered: at most 2 per mille (0.1 % speckle: the sparse march is 1.4x faster than the dense one,
// at 1 % 1.4x slower; MI355X, 4096^2, R = 9).  Unknown counts take the dense march, whose cost does not depend on the map.
// (A map without invalid cells takes the dense kernel too: its clean march is the same code, and a tile with invalid
// cells uploaded later -- tiles are not counted -- is then in safe hands.)
// Which march serves the invalid cells of the resident elevation layer: te_hole_routing.h (plain functions of the upload's
// two counts, shared with the CPU test); the lab switches stay here.
HoleCounts hole_counts(const te_ctx* c) { return HoleCounts{(long long)c->layer_elems, c->invalid_cells, c->invalid_runs}; }
bool clustered_holes(const te_ctx* c) { return holes_clustered(hole_counts(c)); }
bool sparse_holes(const te_ctx* c) {
  static const int force = lab_int("TE_N3_HOLES", 0);  // measurement aid: 1 sparse, 2 dense
  if (force == 1 || force == 2) return force == 1;
  return holes_sparse(hole_counts(c));
}
bool skip_clean_march(const te_ctx* c) {
#ifdef TE_NO_SKIP_CLEAN  // (A/B builds only)
  return false;
#endif
  return sparse_holes(c) && holes_skip_clean_march(hole_counts(c));
}
bool short_strips(const te_ctx* c) {
  static const int force = lab_int("TE_N3_SHORT_STRIPS", -1);  // measurement aid: 0 never, 1 whenever invalid cells were counted
  if (force >= 0) return force == 1 && c->invalid_cells > 0 && !sparse_holes(c);
  return clustered_holes(c) && !sparse_holes(c);
}

// the sparse march's queues; false (and the dense kernel) if the allocation fails
bool ensure_hole_queue(te_ctx* c) {
  if (c->hole_queue) return true;
  if (hipSetDevice(c->device) != hipSuccess) return false;
  void* p = nullptr;
  if (hipMalloc(&p, fast::normals_hole_queue_bytes()) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  c->hole_queue = (char*)p;
  return true;
}

// the step filter's scratch layer at a tie radius (te_fast_step.hip); without it the generic kernels serve
void ensure_tie_scratch(te_ctx* c) {
  if (c->tie_scratch || !c->tables_ready || c->layer_elems == 0) return;
  if (c->cp.step1.n_ties == 0 && c->cp.step2.n_ties == 0) return;
  if (hipSetDevice(c->device) != hipSuccess) return;
  void* p = nullptr;
  if (hipMalloc(&p, c->layer_elems * sizeof(float)) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  c->tie_scratch = (float*)p;
}

void drop_graph(te_ctx* c) {
  for (int k = 0; k < te_ctx::kGraphs; ++k) {
    if (c->graph_exec[k]) (void)hipGraphExecDestroy(c->graph_exec[k]);
    c->graph_exec[k] = nullptr;
  }
}

// Discs of the three footprint checks, SpiralIterator order and clip table of the circular footprint pass.  A failure is
// recorded (fp_tables_rc / fp_tables_err) and reported by the entry points that need the footprint, not by the chain.
int rebuild_footprint_tables_impl(te_ctx* c) {
  const te_params& p = c->params;
  const double res = c->geo.res;
  int rc;
  // ---- circular footprint: discs of the three checks, spiral order, clip table ----------------------
  {
    FootprintParams& f = c->fp;
    if ((rc = build_disc(3.0 * res, res, &f.slope_disc, "footprint slope window"))) return rc;   // TraversabilityMap.cpp:871
    if ((rc = build_disc(2.5 * res, res, &f.step_disc, "footprint step window"))) return rc;     // :798
    f.rmin = p.fp_radius;
    f.rmax = p.fp_radius + p.fp_offset;  // :312 isTraversable(center, radius + offset, ..., radius)
    if ((rc = build_disc(f.rmax, res, &f.fp_disc, "footprint"))) return rc;
    f.def = p.fp_default;
    f.max_gap = p.fp_max_gap;
    f.crit_step = p.fp_critical_step;
    f.check_rough = p.fp_check_roughness;
    {
      const double wr = 3.0 * res, crit_len = p.fp_max_gap / 3.0;
      f.ncrit_slope = (int)floor(2 * wr * crit_len / pow(res, 2));    // :873
      f.ncrit_rough = (int)floor(1.5 * wr * crit_len / pow(res, 2));  // :901
    }
    // SpiralIterator order (grid_map_core): centre, then ring d = 1..nRings, each generated by a perimeter
    // walk from (d, 0) and consumed from the back; only the two outer rings are tested against the circle.
    const Disc& d = f.fp_disc;
    const unsigned nrings = (unsigned)ceil(f.rmax / res);
    std::vector<int16_t> tab;
    auto tie_of = [&](int di, int dj) {
      for (int t = 0; t < d.n_ties; ++t)
        if (d.tie_di[t] == di && d.tie_dj[t] == dj) return true;
      return false;
    };
    auto in_runs = [&](int di, int dj) {
      const int ai = di < 0 ? -di : di, aj = dj < 0 ? -dj : dj;
      return aj <= d.R && d.R >= 0 && d.hw[aj] >= 0 && ai <= d.hw[aj];
    };
    auto push = [&](int di, int dj, bool tie) {
      tab.push_back((int16_t)di);
      tab.push_back((int16_t)dj);
      tab.push_back((int16_t)(int)sqrt((double)(di * di + dj * dj)));  // getCurrentRadius(): integer norm
      tab.push_back((int16_t)(tie ? 1 : 0));
    };
    push(0, 0, false);
    int reach = 0;
    for (unsigned dist = 1; dist <= nrings && dist <= (unsigned)kMaxRadiusCells + 1; ++dist) {
      std::vector<int> ring;
      int px = (int)dist, py = 0;
      do {
        bool keep = true, tie = false;
        if (dist == nrings || dist + 1 == nrings) {
          tie = tie_of(px, py);
          keep = tie || in_runs(px, py);
        }
        if (keep) {
          ring.push_back(px);
          ring.push_back(py);
          ring.push_back(tie ? 1 : 0);
        }
        const int nx = -((py > 0) - (py < 0)), ny = (px > 0) - (px < 0);
        if (nx != 0 && (unsigned)sqrt((double)(px + nx) * (px + nx) + (double)py * py) == dist)
          px += nx;
        else if (ny != 0 && (unsigned)sqrt((double)px * px + (double)(py + ny) * (py + ny)) == dist)
          py += ny;
        else {
          px += nx;
          py += ny;
        }
      } while ((unsigned)px != dist || py != 0);
      for (int k = (int)ring.size() / 3 - 1; k >= 0; --k) {
        push(ring[3 * k], ring[3 * k + 1], ring[3 * k + 2] != 0);
        const int ax = abs(ring[3 * k]), ay = abs(ring[3 * k + 1]);
        reach = ax > reach ? ax : reach;
        reach = ay > reach ? ay : reach;
      }
    }
    f.n_spiral = (int)tab.size() / 4;
    f.reach = reach < 1 ? 1 : reach;
    if (f.reach > 20 || f.n_spiral > kMaxSpiral)
      return fail(TE_ERR_UNSUPPORTED, "footprint radius %.3g m is %d cells; this build supports up to 20", f.rmax, f.reach);
    std::vector<int> ctab((size_t)(2 * f.reach + 1) * (2 * f.reach + 1) * 6);
    fast::build_clip_table(d, f.reach, ctab.data());
    HIP_TRY(hipSetDevice(c->device));
    // the int16 table, followed by the same entries packed into one word each (di | dj << 8 | ring << 16 | tie << 24):
    // the kernels' spiral walk reads those with scalar loads, eight entries at a time
    if (!c->d_spiral) HIP_TRY(hipMalloc((void**)&c->d_spiral, sizeof(int16_t) * 4 * kMaxSpiral + sizeof(uint32_t) * kMaxSpiral));
    std::vector<uint32_t> packed(f.n_spiral);
    for (int k = 0; k < f.n_spiral; ++k)
      packed[k] = ((uint32_t)tab[4 * k] & 0xffu) | (((uint32_t)tab[4 * k + 1] & 0xffu) << 8) | (((uint32_t)tab[4 * k + 2] & 0xffu) << 16) |
                  (((uint32_t)tab[4 * k + 3] & 0xffu) << 24);
    HIP_TRY(hipMemcpyAsync(c->d_spiral + 4 * kMaxSpiral, packed.data(), packed.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    if (!c->fp_clip_table) HIP_TRY(hipMalloc((void**)&c->fp_clip_table, sizeof(int) * (2 * fast::kFpClipInts + kMaxTies)));
    std::vector<int> ctab_full;
    int gen_tab[kMaxTies];
    if (d.n_ties) {  // the disc with the cells on its circle (fixed-point sliding sum of a tie radius, te_footprint4.hip)
      Disc full = d;
      for (int t = 0; t < d.n_ties; ++t) {
        const int ai = abs((int)d.tie_di[t]), aj = abs((int)d.tie_dj[t]);
        if (full.hw[aj] < ai) full.hw[aj] = ai;
        if (full.R < aj) full.R = aj;
      }
      ctab_full.resize(ctab.size());
      fast::build_clip_table(full, f.reach, ctab_full.data());
      HIP_TRY(hipMemcpyAsync(c->fp_clip_table + fast::kFpClipInts, ctab_full.data(), ctab_full.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
      // ... and the offsets on the circle with both parts non-zero, packed (the kernel handles (+-R, 0) and (0, +-R) itself)
      int n_gen = 0;
      for (int t = 0; t < d.n_ties; ++t)
        if (d.tie_di[t] != 0 && d.tie_dj[t] != 0) gen_tab[n_gen++] = ((int)d.tie_di[t] & 0xff) | (((int)d.tie_dj[t] & 0xff) << 8);
      if (n_gen) HIP_TRY(hipMemcpyAsync(c->fp_clip_table + 2 * fast::kFpClipInts, gen_tab, n_gen * sizeof(int), hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(c->d_spiral, tab.data(), tab.size() * sizeof(int16_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->fp_clip_table, ctab.data(), ctab.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  return TE_OK;
}

void rebuild_footprint_tables(te_ctx* c) {
  c->fp_tables_ready = false;
  const int rc = rebuild_footprint_tables_impl(c);
  c->fp_tables_rc = rc;
  if (rc) {
    snprintf(c->fp_tables_err, sizeof(c->fp_tables_err), "%s", g_err);
    (void)hipGetLastError();
  } else {
    c->fp_tables_ready = true;
  }
}

int rebuild_tables(te_ctx* c) {
  c->tables_ready = false;
  c->fp_tables_ready = false;
  drop_graph(c);  // kernel arguments (disc tables, grids) are baked into the captured launches
  if (!c->have_params || !c->have_geo) return TE_OK;
  const te_params& p = c->params;
  const double res = c->geo.res;
  int rc;
  if ((rc = build_disc(p.normals_radius, res, &c->cp.normals, "normals"))) return rc;
  if ((rc = build_disc(p.rough_radius, res, &c->cp.rough, "roughness estimation"))) return rc;
  if ((rc = build_disc(p.step_radius1, res, &c->cp.step1, "step first window"))) return rc;
  if ((rc = build_disc(p.step_radius2, res, &c->cp.step2, "step second window"))) return rc;
  c->cp.same_rough_disc = same_disc(c->cp.normals, c->cp.rough) ? 1 : 0;
  c->cp.axis = p.normals_axis;
  c->cp.slope_crit = p.slope_critical;
  c->cp.step_crit = p.step_critical;
  c->cp.rough_crit = p.rough_critical;
  c->cp.step_ncrit = p.step_ncrit;
  c->cp.w_scale = p.w_scale;
  c->cp.w_slope = p.w_slope;
  c->cp.w_step = p.w_step;
  c->cp.w_rough = p.w_rough;
  // x/y moments of the normals disc clipped by the map border, for the sliding-disc kernel
  // (a tie radius: the table of the disc WITH the cells on its circle behind it, then the circle's offsets with both
  // parts non-zero -- te_normals3.hip, TIES march)
  if (c->cp.normals.R >= 1 || c->cp.normals.n_ties != 0) {
    const Disc& dn = c->cp.normals;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->clip_table) HIP_TRY(hipMalloc((void**)&c->clip_table, sizeof(int) * (2 * fast::kClipInts + kMaxTies)));
    std::vector<int> tab, tab_full;
    int gen_tab[kMaxTies];
    if (dn.n_ties == 0) {
      const int R = dn.R;
      tab.resize((size_t)(2 * R + 1) * (2 * R + 1) * 6);
      fast::build_clip_table(dn, R, tab.data());
      HIP_TRY(hipMemcpyAsync(c->clip_table, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    } else {
      Disc full = dn;
      int n_gen = 0;
      for (int t = 0; t < dn.n_ties; ++t) {
        const int ai = abs((int)dn.tie_di[t]), aj = abs((int)dn.tie_dj[t]);
        if (full.hw[aj] < ai) full.hw[aj] = ai;
        if (full.R < aj) full.R = aj;
        if (dn.tie_di[t] != 0 && dn.tie_dj[t] != 0) gen_tab[n_gen++] = ((int)dn.tie_di[t] & 0xff) | (((int)dn.tie_dj[t] & 0xff) << 8);
      }
      const int R = dn.reach;
      tab_full.resize((size_t)(2 * R + 1) * (2 * R + 1) * 6);
      fast::build_clip_table(full, R, tab_full.data());
      HIP_TRY(hipMemcpyAsync(c->clip_table + fast::kClipInts, tab_full.data(), tab_full.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
      if (n_gen) HIP_TRY(hipMemcpyAsync(c->clip_table + 2 * fast::kClipInts, gen_tab, n_gen * sizeof(int), hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  c->L.clip_table = c->clip_table;

  c->tables_ready = true;
  rebuild_footprint_tables(c);
  return TE_OK;
}

void free_layers(te_ctx* c) {
  drop_graph(c);
  if (c->slab) (void)hipFree(c->slab);
  c->slab = nullptr;
  guard_cache_generation().fetch_add(1, std::memory_order_acq_rel);  // (layer_has_guard_rows: verdicts about freed memory)
  if (c->poly_x) (void)hipFree(c->poly_x);
  c->poly_x = c->poly_rot = nullptr;
  if (c->poly_stream) (void)hipFree(c->poly_stream);
  c->poly_stream = nullptr;
  c->poly_stream_cap = 0;
  if (c->robot_slope) (void)hipFree(c->robot_slope);
  c->robot_slope = nullptr;
  if (c->tie_scratch) (void)hipFree(c->tie_scratch);
  c->tie_scratch = nullptr;
  c->have_robot_slope = false;
  memset(&c->L, 0, sizeof(c->L));
  c->layer_elems = 0;
  c->trav_ptr_out = false;
  c->have_elev = false;
  c->chain_done = false;
  c->footprint_done = false;
}

float* layer_ptr(te_ctx* c, int layer) {
  switch (layer) {
    case TE_LAYER_ELEVATION: return c->L.elev;
    case TE_LAYER_SLOPE: return c->L.slope;
    case TE_LAYER_STEP: return c->L.step;
    case TE_LAYER_ROUGHNESS: return c->L.rough;
    case TE_LAYER_TRAVERSABILITY: return c->L.trav;
    case TE_LAYER_FOOTPRINT: return c->L.footprint;
    case TE_LAYER_NORMAL_X: return c->L.nx;
    case TE_LAYER_NORMAL_Y: return c->L.ny;
    case TE_LAYER_NORMAL_Z: return c->L.nz;
    case TE_LAYER_SLOPE_FOOTPRINT: return c->L.slope_fp;
    case TE_LAYER_STEP_FOOTPRINT: return c->L.step_fp;
    case TE_LAYER_ROUGHNESS_FOOTPRINT: return c->L.rough_fp;
    case TE_LAYER_TRAVERSABILITY_X: return c->poly_x;
    case TE_LAYER_TRAVERSABILITY_ROT: return c->poly_rot;
    case TE_LAYER_ROBOT_SLOPE: return c->robot_slope;
    default: return nullptr;
  }
}

// the optional input layer robot_slope exists from its first upload on (every cell NaN = not valid until written)
int ensure_input_layer(te_ctx* c, int layer) {
  if (layer != TE_LAYER_ROBOT_SLOPE || c->robot_slope) return TE_OK;
  HIP_TRY(hipSetDevice(c->device));
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, c->layer_elems * sizeof(float)));
  const hipError_t e = hipMemsetD32Async((hipDeviceptr_t)p, 0x7fc00000, c->layer_elems, c->stream);
  if (e != hipSuccess) {
    (void)hipFree(p);
    return fail(TE_ERR_HIP, "robot_slope layer: %s", hipGetErrorString(e));
  }
  c->robot_slope = (float*)p;
  return TE_OK;
}

int run_chain_locked(te_ctx* c, unsigned flags, const Region& r) {
  if (!c->have_params || !c->have_geo) return fail(TE_ERR_NOT_READY, "te_run_chain: set params and geometry first");
  if (!c->tables_ready) {
    int rc = rebuild_tables(c);
    if (rc) return rc;
  }
  if (!c->have_elev) return fail(TE_ERR_NOT_READY, "te_run_chain: no elevation uploaded");
  HIP_TRY(hipSetDevice(c->device));
  // Two streams (step filter || normals kernel) pay from about 2^21 cells: below that the launch is a handful of
  // short kernels and the fork / join events cost more than the overlap gains -- one stream, the combine fused into the
  // normals kernel (MI355X, R = 5, chain: 256^2 0.045 -> 0.030 ms, 512^2 0.042 -> 0.032, 1024^2 0.052 -> 0.046, 2048^2 equal).
  static const bool force_two = lab_flag("TE_TWO_STREAMS");  // measurement aid
  const bool small = !force_two && (size_t)c->geo.rows * c->geo.cols * c->geo.batch < ((size_t)1 << 21);
  c->L.aux_stream = ((flags & TE_RUN_SEQUENTIAL) || small) ? nullptr : c->aux_stream;
  // whole-map run with the footprint pass right behind: the mask kernel writes the combined layer
  if (flags & TE_RUN_NORMALS_ONLY) flags &= ~(TE_RUN_FOOTPRINT | TE_RUN_FOOTPRINT_MEMO);
  // (only if the footprint pass can run at all: with a footprint this build cannot handle the combined layer would
  // never be written)
  if ((flags & TE_RUN_FOOTPRINT) && !c->fp_tables_ready)
    return fail(c->fp_tables_rc ? c->fp_tables_rc : TE_ERR_NOT_READY, "%s", c->fp_tables_err[0] ? c->fp_tables_err : "footprint tables not built");
  c->combine_deferred = (r.map < 0) && !(flags & TE_RUN_SEQUENTIAL) && (flags & TE_RUN_FOOTPRINT);
  if (c->combine_deferred) flags |= kDeferCombine;
  c->L.ev_fork = c->ev_fork;
  c->L.ev_join = c->ev_join;
  c->L.fb_walk = c->opt_fb_walk;
  c->L.fb_blocks_per_cu = c->opt_fb_blocks_per_cu;
  c->L.sparse_holes = sparse_holes(c) && ensure_hole_queue(c) ? 1 : 0;  // (run_whole_locked allocates before it captures)
  c->L.no_holes = c->invalid_cells == 0 ? 1 : 0;
  c->L.skip_clean = c->L.sparse_holes && skip_clean_march(c) ? 1 : 0;
  c->L.short_strips = short_strips(c) ? 1 : 0;
  c->L.hole_queue = c->hole_queue;
  ensure_tie_scratch(c);  // (likewise)
  c->L.tie_scratch = c->tie_scratch;
  c->cp.rank_rule = c->opt_rank_rule;
  if (c->opt_rank_rule) flags |= TE_RUN_GENERIC_KERNELS;  // (the rule lives in the generic normals kernel only)
  HIP_TRY(launch_chain(c->geo, c->cp, c->L, r, flags, c->stream));
  c->chain_done = true;
  c->footprint_done = false;  // the layers the footprint pass reads have changed
  if (r.map < 0) c->trav_external = false;  // every cell of the combined layer now comes from the chain
  return TE_OK;
}

// fresh: called right behind a whole-map chain in the same entry point (the combined layer is the chain's, cell for cell)
int run_footprint_locked(te_ctx* c, unsigned flags, bool fresh = false) {
  if (!c->chain_done || !c->tables_ready)
    return fail(TE_ERR_NOT_READY, "te_run_footprint: run the filter chain first (it produces the layers the footprint reads)");
  if (!c->fp_tables_ready) return fail(c->fp_tables_rc ? c->fp_tables_rc : TE_ERR_NOT_READY, "%s", c->fp_tables_err);
  HIP_TRY(hipSetDevice(c->device));
  // bound of the combined layer, if the chain wrote it: scores lie in [0, 1], so w_scale * (w_slope + w_step + w_rough)
  const ChainParams& q = c->cp;
  const bool bounded = !c->trav_external && (fresh || !c->trav_ptr_out) && q.w_scale >= 0.0f && q.w_slope >= 0.0f && q.w_step >= 0.0f && q.w_rough >= 0.0f;
  const double trav_cap = bounded ? (double)q.w_scale * ((double)q.w_slope + (double)q.w_step + (double)q.w_rough) : -1.0;
  HIP_TRY(launch_footprint(c->geo, c->fp, c->L, c->d_spiral, c->fp_clip_table, (flags & TE_RUN_FOOTPRINT_MEMO) != 0,
                           c->combine_deferred ? &c->cp : nullptr, trav_cap, c->stream));
  c->combine_deferred = false;
  c->footprint_done = true;
  return TE_OK;
}

// Whole-map chain (+ footprint): the launch sequence is captured once per (flags, hole hints) into a hipGraph and replayed
// from 2^22 cells on (TE_OPT_GRAPH_REPLAY: always / never).  Measured on MI355X / ROCm 7.2, direct -> replayed, us per launch
// (tools/lab/graph_small_ab.py, round 6): bag map 17 -> 23, 256^2 25 -> 31, 1024^2 45 -> 52 (the replay costs 6 us on a
// launch that is a handful of short kernels on one stream), 2048^2 85 -> 79 and 121 -> 118 with the footprint pass, 4096^2
// 17 us saved (round 1).  Any capture problem switches the context back to direct launches for good.
int run_whole_locked(te_ctx* c, unsigned flags) {
  const Region r = {-1, 0, 0, c->geo.rows, c->geo.cols};
  static const bool no_graph = lab_flag("TE_NO_GRAPH");
  const bool large = c->opt_graph == 1 || (c->opt_graph == 0 && (size_t)c->geo.rows * c->geo.cols * c->geo.batch >= ((size_t)1 << 22));
  // (only the defined TE_RUN_* bits: the graph key below puts its own hints into the upper bits of the same word)
  flags &= TE_RUN_KEEP_NORMALS | TE_RUN_FOOTPRINT | TE_RUN_GENERIC_KERNELS | TE_RUN_FOOTPRINT_MEMO | TE_RUN_SEQUENTIAL | TE_RUN_NORMALS_ONLY;
  if (c->opt_rank_rule) flags |= TE_RUN_GENERIC_KERNELS;
  if (flags & TE_RUN_NORMALS_ONLY) flags &= ~(TE_RUN_FOOTPRINT | TE_RUN_FOOTPRINT_MEMO);
  if (!no_graph && large && !(flags & TE_RUN_NORMALS_ONLY) && c->graph_ok && c->have_params && c->have_geo && c->have_elev) {
    if (!c->tables_ready) {
      int rc = rebuild_tables(c);
      if (rc) return rc;
    }
    HIP_TRY(hipSetDevice(c->device));
    int slot = -1;
    // (the captured launches bake in which k_normals3 variant runs: the hint is part of the key)
    ensure_tie_scratch(c);
    const unsigned key = flags | (sparse_holes(c) && ensure_hole_queue(c) ? 0x80000000u : 0u) | (c->invalid_cells == 0 ? 0x40000000u : 0u) | (skip_clean_march(c) ? 0x20000000u : 0u) | (short_strips(c) ? 0x10000000u : 0u);
    for (int k = 0; k < te_ctx::kGraphs; ++k)
      if (c->graph_exec[k] && c->graph_flags[k] == key) slot = k;
    if (slot < 0) {
      slot = c->graph_next;
      c->graph_next = (c->graph_next + 1) % te_ctx::kGraphs;
      if (c->graph_exec[slot]) (void)hipGraphExecDestroy(c->graph_exec[slot]);
      c->graph_exec[slot] = nullptr;
      hipGraph_t graph = nullptr;
      int rc = TE_OK;
      hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed);
      if (e == hipSuccess) {
        rc = run_chain_locked(c, flags, r);
        if (!rc && (flags & TE_RUN_FOOTPRINT)) rc = run_footprint_locked(c, flags, true);
        e = hipStreamEndCapture(c->stream, &graph);
      }
      if (e == hipSuccess && rc == TE_OK && graph) e = hipGraphInstantiate(&c->graph_exec[slot], graph, nullptr, nullptr, 0);
      if (graph) (void)hipGraphDestroy(graph);
      if (e != hipSuccess || rc != TE_OK || !c->graph_exec[slot]) {
        (void)hipGetLastError();
        drop_graph(c);
        c->graph_ok = false;
        slot = -1;
      } else {
        c->graph_flags[slot] = key;
      }
    }
    if (slot >= 0) {
      TraceRange tr("te_run_chain: graph replay (chain + footprint kernels)");
      HIP_TRY(hipGraphLaunch(c->graph_exec[slot], c->stream));
      c->trav_external = false;  // (as run_chain_locked: every cell of the combined layer now comes from the chain)
      c->chain_done = true;
      c->footprint_done = (flags & TE_RUN_FOOTPRINT) != 0;
      c->combine_deferred = false;
      return TE_OK;
    }
  }
  int rc = run_chain_locked(c, flags, r);
  if (!rc && (flags & TE_RUN_FOOTPRINT)) rc = run_footprint_locked(c, flags, true);
  return rc;
}

int sync_tiles(te_ctx* c) {  // the copy streams of the streaming-tile calls
  if (!c->tiles_pending) return TE_OK;
  if (c->in_stream) HIP_TRY(hipStreamSynchronize(c->in_stream));
  if (c->out_stream) HIP_TRY(hipStreamSynchronize(c->out_stream));
  c->tiles_pending = false;
  return TE_OK;
}

}  // namespace shim
}  // namespace te

extern "C" {

const char* te_last_error(void) { return g_err; }
const char* te_version(void) { return "travgpu 0.1 (gfx950)"; }

int te_params_default(te_params* p) {
  if (!p) return fail(TE_ERR_INVALID_ARG, "te_params_default: NULL");
  memset(p, 0, sizeof(*p));
  p->size = (uint32_t)sizeof(te_params);
  p->abi_version = TE_ABI_VERSION;
  // traversability_estimation/config/robot_filter_parameter.yaml:3-33
  p->normals_radius = 0.05;
  p->normals_axis = 2;
  p->slope_critical = 1.0;
  p->step_critical = 0.12;
  p->step_radius1 = 0.04;
  p->step_radius2 = 0.04;
  p->step_ncrit = 4;
  p->rough_critical = 0.05;
  p->rough_radius = 0.05;
  p->w_scale = 1.0f / 3.0f;
  p->w_slope = p->w_step = p->w_rough = 1.0f;
  // robot_footprint_parameter.yaml:4-8, robot.yaml:10, TraversabilityMap.cpp:117-126
  p->fp_radius = 0.30;
  p->fp_offset = 0.15;
  p->fp_default = 0.3;
  p->fp_max_gap = 0.3;
  p->fp_critical_step = 0.12;
  p->fp_check_roughness = 0;
  return TE_OK;
}

int te_params_validate(const te_params* p) {
  if (!p) return fail(TE_ERR_INVALID_ARG, "te_params: NULL");
  if (p->size != sizeof(te_params) || p->abi_version != TE_ABI_VERSION)
    return fail(TE_ERR_INVALID_ARG, "te_params: size/abi mismatch (got %u/%u, want %zu/%d)", p->size, p->abi_version,
                sizeof(te_params), TE_ABI_VERSION);
  // same messages as the reference's configure()s
  if (!(p->slope_critical <= M_PI_2 && p->slope_critical >= 0.0))  // SlopeFilter.cpp:41
    return fail(TE_ERR_BAD_PARAM, "Critical slope must be in the interval [0, PI/2]");
  if (!(p->step_critical >= 0.0))  // StepFilter.cpp:45
    return fail(TE_ERR_BAD_PARAM, "Critical step height must be greater than zero.");
  if (!(p->step_radius1 >= 0.0))  // :58
    return fail(TE_ERR_BAD_PARAM, "'first_window_radius' must be greater than zero.");
  if (!(p->step_radius2 >= 0.0))  // :71
    return fail(TE_ERR_BAD_PARAM, "'second_window_radius' must be greater than zero.");
  if (p->step_ncrit <= 0)  // :84
    return fail(TE_ERR_BAD_PARAM, "'critical_cell_number' must be greater than zero.");
  if (!(p->rough_critical >= 0.0))  // RoughnessFilter.cpp:43
    return fail(TE_ERR_BAD_PARAM, "Critical roughness must be greater than zero");
  if (!(p->rough_radius >= 0.0))  // :55
    return fail(TE_ERR_BAD_PARAM, "Roughness estimation radius must be greater than zero");
  if (!(p->normals_radius >= 0.0)) return fail(TE_ERR_BAD_PARAM, "normals radius must not be negative");
  if (p->normals_axis < 0 || p->normals_axis > 2)
    return fail(TE_ERR_BAD_PARAM, "normal_vector_positive_axis must be x, y or z");
  if (!(p->fp_radius >= 0.0) || !(p->fp_offset >= 0.0))
    return fail(TE_ERR_BAD_PARAM, "footprint radius/offset must not be negative");
  if (!(p->fp_max_gap >= 0.0)) return fail(TE_ERR_BAD_PARAM, "max_gap_width must not be negative");
  return TE_OK;
}

int te_device_count(int* count) {
  if (!count) return fail(TE_ERR_INVALID_ARG, "te_device_count: NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(TE_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return TE_OK;
}

int te_create(int device, te_ctx** out) {
  if (!out) return fail(TE_ERR_INVALID_ARG, "te_create: NULL out");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(TE_ERR_NO_DEVICE, "te_create: no HIP device visible (libtravgpu has no CPU fallback)");
  if (device < 0 || device >= n) return fail(TE_ERR_INVALID_ARG, "te_create: device %d of %d", device, n);
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(TE_ERR_NO_DEVICE, "te_create: device %d is %s; this library is built for gfx950 only", device,
                prop.gcnArchName);
  te_ctx* c = new (std::nothrow) te_ctx();
  if (!c) return fail(TE_ERR_INVALID_ARG, "te_create: out of host memory");
  c->device = device;
  memset(&c->L, 0, sizeof(c->L));
  memset(&c->geo, 0, sizeof(c->geo));
  memset(&c->cp, 0, sizeof(c->cp));
  memset(&c->fp, 0, sizeof(c->fp));
  te_params_default(&c->params);
  c->have_params = true;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&c->ev0);
  if (e == hipSuccess) e = hipEventCreate(&c->ev1);
  if (e == hipSuccess) {  // the (shorter) step-filter kernels go first when both streams have work
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    e = hipStreamCreateWithPriority(&c->aux_stream, hipStreamNonBlocking, hi);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
  if (e != hipSuccess) {
    delete c;
    return fail(TE_ERR_HIP, "te_create: %s", hipGetErrorString(e));
  }
  *out = c;
  return TE_OK;
}

int te_destroy(te_ctx* c) {
  if (!c) return TE_OK;
  {
    CtxLock lk(c);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->prefetch_thread.joinable()) {
      {
        std::lock_guard<std::mutex> pl(c->pf_mu);
        c->pf_quit = true;
      }
      c->pf_cv.notify_all();
      c->prefetch_thread.join();
    }
    free_layers(c);
    c->stager.release();
    c->prefetcher.release();
    if (c->prefetch_order) (void)hipStreamDestroy(c->prefetch_order);
    if (c->d_spiral) (void)hipFree(c->d_spiral);
    if (c->d_count) (void)hipFree(c->d_count);
    if (c->hole_queue) (void)hipFree(c->hole_queue);
    if (c->clip_table) (void)hipFree(c->clip_table);
    if (c->fp_clip_table) (void)hipFree(c->fp_clip_table);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    for (hipStream_t st : {c->in_stream, c->out_stream})
      if (st) {
        (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
      }
    for (te_ctx::TileSlot* sl : {&c->in_slot[0], &c->in_slot[1], &c->out_slot[0], &c->out_slot[1]}) {
      if (sl->buf) (void)hipFree(sl->buf);
      if (sl->ready) (void)hipEventDestroy(sl->ready);
      if (sl->freed) (void)hipEventDestroy(sl->freed);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
  }
  delete c;
  return TE_OK;
}

int te_set_params(te_ctx* c, const te_params* p) {
  if (!c) return fail(TE_ERR_INVALID_ARG, "te_set_params: NULL ctx");
  int rc = te_params_validate(p);
  if (rc) return rc;
  CtxLock lk(c, /*beside_prefetch*/ true);
  te_params old = c->params;
  c->params = *p;
  c->have_params = true;
  rc = rebuild_tables(c);
  if (rc) {
    c->params = old;
    (void)rebuild_tables(c);
    return rc;
  }
  // the filter layers stay valid when only the footprint part (fp_*, the tail of the struct) changed:
  // traversabilityFootprint(radius, offset) is called with a new radius on an unchanged map
  if (memcmp(&old, p, offsetof(te_params, fp_radius)) != 0) c->chain_done = false;
  c->footprint_done = false;
  return TE_OK;
}

int te_set_option(te_ctx* c, int option, int value) {
  if (!c) return fail(TE_ERR_INVALID_ARG, "te_set_option: NULL ctx");
  CtxLock lk(c);
  switch (option) {
    case TE_OPT_FP_BLOCKED_WALK:
      if (value < 0 || value > 2) return fail(TE_ERR_INVALID_ARG, "te_set_option: TE_OPT_FP_BLOCKED_WALK takes 0 (by list length), 1 (per wavefront), 2 (per lane)");
      c->opt_fb_walk = value;
      break;
    case TE_OPT_FP_BLOCKED_BLOCKS_PER_CU:
      if (value < 0 || value > 32) return fail(TE_ERR_INVALID_ARG, "te_set_option: TE_OPT_FP_BLOCKED_BLOCKS_PER_CU takes 0 (default) .. 32");
      c->opt_fb_blocks_per_cu = value;
      break;
    case TE_OPT_POLYGON_PER_CELL:
      c->opt_polygon_per_cell = value != 0;
      break;
    case TE_OPT_GRAPH_REPLAY:
      if (value < 0 || value > 2) return fail(TE_ERR_INVALID_ARG, "te_set_option: TE_OPT_GRAPH_REPLAY takes 0 (by size), 1 (always), 2 (never)");
      c->opt_graph = value;
      break;
    case TE_OPT_BCAST_RCCL:
      c->opt_bcast_rccl = value != 0;
      return TE_OK;  // (no launch depends on it)
    case TE_OPT_NORMALS_RANK_RULE:
      c->opt_rank_rule = value != 0;
      c->chain_done = false;
      c->footprint_done = false;
      break;
    default:
      return fail(TE_ERR_INVALID_ARG, "te_set_option: unknown option %d", option);
  }
  drop_graph(c);  // (captured launches bake the choice in)
  return TE_OK;
}

int te_get_params(te_ctx* c, te_params* p) {
  if (!c || !p) return fail(TE_ERR_INVALID_ARG, "te_get_params: NULL");
  CtxLock lk(c, /*beside_prefetch*/ true);
  *p = c->params;
  return TE_OK;
}

int te_set_geometry(te_ctx* c, int rows, int cols, int batch, double res, double pos_x, double pos_y) {
  if (!c) return fail(TE_ERR_INVALID_ARG, "te_set_geometry: NULL ctx");
  if (rows <= 0 || cols <= 0 || batch <= 0 || !(res > 0.0) || !isfinite(res) || !isfinite(pos_x) || !isfinite(pos_y))
    return fail(TE_ERR_INVALID_ARG, "te_set_geometry: rows=%d cols=%d batch=%d res=%g", rows, cols, batch, res);
  CtxLock lk(c);
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const size_t elems = (size_t)rows * cols * batch;
  // a new shape always gets a fresh slab: the fix-up flag array is sized from (rows, cols, batch), not from the cell
  // count (a transposed map of equal size needs a different number of 64x16 tiles), and layers computed for another
  // shape must not be served as if they belonged to this one
  if (elems != c->layer_elems || rows != c->geo.rows || cols != c->geo.cols || batch != c->geo.batch) {
    free_layers(c);
    // one slab: 13 float layers + 1 byte layer, each 256-byte aligned
    const size_t lb = (elems * sizeof(float) + 255) & ~(size_t)255;
    const size_t ub = (elems + 255) & ~(size_t)255;
    void* slab = nullptr;
    Geo gtmp;
    gtmp.rows = rows;
    gtmp.cols = cols;
    gtmp.batch = batch;
    const size_t fb = ((size_t)fast::normals_fast_max_blocks(gtmp) * sizeof(int) + 255) & ~(size_t)255;
    // (+ the footprint pass's list of cells with an untraversable cell in their disc: one 32-bit entry per cell at
    // most, and its counter)
    const size_t list_cap = elems + fast::f4_list_slack(rows, cols, batch);
    const size_t qb = (list_cap * sizeof(unsigned) + 255) & ~(size_t)255;
    // (guard: kSlabGuardRows rows of slack before the first and behind the last layer -- te_internal.h)
    const size_t guard = ((size_t)kSlabGuardRows * (size_t)rows * sizeof(float) + 255) & ~(size_t)255;
    // (+ one byte per 64 x 4 cells: "holds an untraversable cell", written by the mask kernel, 1 = unknown until then)
    const size_t ufb = (untrav_flag_bytes(rows, cols, batch) + 255) & ~(size_t)255;
    // (+ the sum kernel's scratch: a second array of the list's size, see Layers::fp_scratch)
    const size_t pcb = qb;
    const size_t total = guard + 13 * lb + ub + fb + qb + 256 + ufb + pcb + guard;
    hipError_t e = hipMalloc(&slab, total);
    if (e != hipSuccess) return fail(TE_ERR_HIP, "te_set_geometry: hipMalloc(%zu bytes): %s", total, hipGetErrorString(e));
    c->slab = slab;
    char* b = (char*)slab + guard;
    float** ptrs[13] = {&c->L.elev, &c->L.slope, &c->L.step,     &c->L.rough,   &c->L.trav,     &c->L.footprint, &c->L.nx,
                        &c->L.ny,   &c->L.nz,    &c->L.slope_fp, &c->L.step_fp, &c->L.rough_fp, &c->L.step_height};
    for (int k = 0; k < 13; ++k) *ptrs[k] = (float*)(b + (size_t)k * lb);
    c->L.untrav = (uint8_t*)(b + 13 * lb);
    c->L.block_flags = (int*)(b + 13 * lb + ub);
    c->L.fp_blocked = list_cap < ((size_t)1 << 32) ? (unsigned*)(b + 13 * lb + ub + fb) : nullptr;
    c->L.fp_blocked_count = (unsigned*)(b + 13 * lb + ub + fb + qb);
    c->L.fp_blocked_cap = list_cap;
    c->L.untrav_flags = (uint8_t*)(b + 13 * lb + ub + fb + qb + 256);
    c->L.fp_scratch = list_cap < ((size_t)1 << 32) ? (unsigned*)(b + 13 * lb + ub + fb + qb + 256 + ufb) : nullptr;
    c->layer_elems = elems;
    // outputs read as NaN until computed, like GridMap::add()
    HIP_TRY(hipMemsetAsync(slab, 0xFF, guard + 13 * lb + ub + fb, c->stream));
    HIP_TRY(hipMemsetAsync(b + 13 * lb + ub + fb + qb + 256 + ufb + pcb, 0xFF, guard, c->stream));
    HIP_TRY(hipMemsetAsync(c->L.untrav_flags, 0x01, ufb, c->stream));
    // the mask layer holds 0 / 1 only (k_fp_slide5 packs the byte as it is): "untraversable" until the mask kernel has
    // looked at the cell, as a byte of 0xFF would also say -- but 1 stays inside the packed word's flag bit
    HIP_TRY(hipMemsetAsync(c->L.untrav, 0x01, ub, c->stream));
    HIP_TRY(hipMemsetAsync(c->L.fp_blocked_count, 0, 256, c->stream));
    // the fix-up flags are zero between launches: k_normals_fixup clears every flag it consumes
    HIP_TRY(hipMemsetAsync(c->L.block_flags, 0, fb, c->stream));
  }
  c->geo.rows = rows;
  c->geo.cols = cols;
  c->geo.batch = batch;
  c->geo.res = res;
  c->geo.len_x = (double)rows * res;  // GridMap::setGeometry: length = size * resolution
  c->geo.len_y = (double)cols * res;
  c->geo.pos_x = pos_x;
  c->geo.pos_y = pos_y;
  c->geo.ax = pos_x + (0.5 * c->geo.len_x - 0.5 * res);
  c->geo.ay = pos_y + (0.5 * c->geo.len_y - 0.5 * res);
  c->have_geo = true;
  c->chain_done = false;
  c->footprint_done = false;
  return rebuild_tables(c);
}

int te_run_filter(te_ctx* c, int filter, unsigned flags) {
  if (!c) return fail(TE_ERR_INVALID_ARG, "te_run_filter: NULL ctx");
  static const char* const kFilterRange[] = {"te_run_filter", "te_run_filter: slope", "te_run_filter: step", "te_run_filter: roughness",
                                             "te_run_filter: combine", "te_run_filter: normals"};
  TraceRange tr(kFilterRange[(filter >= 0 && filter < 6) ? filter : 0]);
  CtxLock lk(c, /*beside_prefetch*/ true, filter_layers(filter));
  if (!c->have_geo || !c->have_params) return fail(TE_ERR_NOT_READY, "te_run_filter: set params and geometry first");
  if (!c->tables_ready) {
    int rc = rebuild_tables(c);
    if (rc) return rc;
  }
  if (filter < TE_FILTER_SLOPE || filter > TE_FILTER_NORMALS) return fail(TE_ERR_INVALID_ARG, "te_run_filter: bad filter %d", filter);
  if ((filter == TE_FILTER_STEP || filter == TE_FILTER_ROUGHNESS || filter == TE_FILTER_NORMALS) && !c->have_elev)
    return fail(TE_ERR_NOT_READY, "te_run_filter: no elevation uploaded");
  HIP_TRY(hipSetDevice(c->device));
  ensure_tie_scratch(c);
  c->L.tie_scratch = c->tie_scratch;
  // (the hints the normals kernel takes from the upload's count, as run_chain_locked sets them: never stale ones)
  c->L.sparse_holes = sparse_holes(c) && ensure_hole_queue(c) ? 1 : 0;
  c->L.hole_queue = c->hole_queue;
  c->L.no_holes = c->invalid_cells == 0 ? 1 : 0;
  c->L.skip_clean = c->L.sparse_holes && skip_clean_march(c) ? 1 : 0;
  c->L.short_strips = short_strips(c) ? 1 : 0;
  c->cp.rank_rule = c->opt_rank_rule;  // (TE_OPT_NORMALS_RANK_RULE: as in the chain)
  if (c->opt_rank_rule) flags |= TE_RUN_GENERIC_KERNELS;
  HIP_TRY(launch_filter(c->geo, c->cp, c->L, filter, flags, c->stream));
  // A single plugin's filter overwrites score layers from whatever inputs are resident (TE_FILTER_NORMALS also slope
  // and roughness, with the normals radius): the layers no longer form one chain result, so region re-filters, the
  // footprint pass and the path checks must not build on them.
  c->chain_done = false;
  c->footprint_done = false;
  c->trav_external = true;
  return TE_OK;
}

int te_run_chain(te_ctx* c, unsigned flags) {
  if (!c) return fail(TE_ERR_INVALID_ARG, "te_run_chain: NULL ctx");
  TraceRange tr("te_run_chain");
  CtxLock lk(c);
  if (!c->have_geo) return fail(TE_ERR_NOT_READY, "te_run_chain: geometry not set");
  return run_whole_locked(c, flags);
}

int te_run_chain_region(te_ctx* c, unsigned flags, int map, int row0, int col0, int h, int w) {
  if (!c) return fail(TE_ERR_INVALID_ARG, "te_run_chain_region: NULL ctx");
  TraceRange tr("te_run_chain_region");
  CtxLock lk(c);
  if (!c->have_geo) return fail(TE_ERR_NOT_READY, "te_run_chain_region: geometry not set");
  if (map < 0 || map >= c->geo.batch || row0 < 0 || col0 < 0 || h <= 0 || w <= 0 || row0 + h > c->geo.rows ||
      col0 + w > c->geo.cols)
    return fail(TE_ERR_INVALID_ARG, "te_run_chain_region: rectangle outside the map");
  if (!c->chain_done) return fail(TE_ERR_NOT_READY, "te_run_chain_region: run the full chain once first");
  const Region r = {map, row0, col0, row0 + h, col0 + w};
  const bool want_fp = (flags & (TE_RUN_FOOTPRINT | TE_RUN_FOOTPRINT_MEMO)) != 0;
  const bool fp_was_done = c->footprint_done;
  if (want_fp && !fp_was_done)
    return fail(TE_ERR_NOT_READY, "te_run_chain_region: the footprint flag refreshes a complete traversability_footprint layer; run the "
                                  "whole-map footprint pass once first (te_run_chain with TE_RUN_FOOTPRINT, or te_run_footprint)");
  int rc = run_chain_locked(c, flags & ~(TE_RUN_FOOTPRINT | TE_RUN_FOOTPRINT_MEMO), r);
  if (rc || !want_fp) return rc;
  // The scores changed within the chain's reach of the rectangle (launch_chain re-filters and re-combines exactly that);
  // the footprint pass follows on the cells that can see them.
  // (checkForStep also follows a ray of up to max_gap_width and a Bresenham line along it through the ELEVATION, which
  // changed in the rectangle itself: the mask of a cell depends on elevations up to 2.5 + 1 + max_gap/res cells away; the
  // mask is recomputed within 3 cells of `changed`)
  const int gap_cells = (int)ceil(c->params.fp_max_gap / c->geo.res) + 5;
  const int reach = chain_max_reach(c->cp);
  const int grow = reach > gap_cells - 3 ? reach : gap_cells - 3;
  Region changed = r;
  changed.i0 = r.i0 - grow < 0 ? 0 : r.i0 - grow;
  changed.j0 = r.j0 - grow < 0 ? 0 : r.j0 - grow;
  changed.i1 = r.i1 + grow > c->geo.rows ? c->geo.rows : r.i1 + grow;
  changed.j1 = r.j1 + grow > c->geo.cols ? c->geo.cols : r.j1 + grow;
  const ChainParams& q = c->cp;
  const bool bounded = !c->trav_external && !c->trav_ptr_out && q.w_scale >= 0.0f && q.w_slope >= 0.0f && q.w_step >= 0.0f && q.w_rough >= 0.0f;
  const double trav_cap = bounded ? (double)q.w_scale * ((double)q.w_slope + (double)q.w_step + (double)q.w_rough) : -1.0;
  HIP_TRY(launch_footprint(c->geo, c->fp, c->L, c->d_spiral, c->fp_clip_table, (flags & TE_RUN_FOOTPRINT_MEMO) != 0, nullptr, trav_cap,
                           c->stream, &changed));
  c->footprint_done = true;  // complete before, refreshed where it could change
  return TE_OK;
}

int te_run_footprint(te_ctx* c) {
  if (!c) return fail(TE_ERR_INVALID_ARG, "te_run_footprint: NULL ctx");
  TraceRange tr("te_run_footprint");
  CtxLock lk(c);
  return run_footprint_locked(c, TE_RUN_FOOTPRINT_MEMO);
}

int te_sync(te_ctx* c) {
  if (!c) return fail(TE_ERR_INVALID_ARG, "te_sync: NULL ctx");
  CtxLock lk(c);
  HIP_TRY(hipSetDevice(c->device));
  // A blocking hipStreamSynchronize parks the thread and is woken by an interrupt; for the launches of this library
  // (a few hundred microseconds) that wake-up is a visible part of the latency, so the stream is polled first
  // (TE_SYNC_SPIN_US microseconds, default 2000; 0: block at once).
  static const long spin_us = lab_int("TE_SYNC_SPIN_US", 2000);
  if (spin_us > 0) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t e = hipStreamQuery(c->stream);
      if (e == hipSuccess) return sync_tiles(c);
      if (e != hipErrorNotReady) return fail(TE_ERR_HIP, "te_sync: %s", hipGetErrorString(e));
      if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
    }
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return sync_tiles(c);
}

int te_time_chain(te_ctx* c, unsigned flags, int warmup, int iters, float* ms_per_iter) {
  if (!c || !ms_per_iter || iters <= 0 || warmup < 0) return fail(TE_ERR_INVALID_ARG, "te_time_chain: bad argument");
  CtxLock lk(c);
  if (!c->have_geo) return fail(TE_ERR_NOT_READY, "te_time_chain: geometry not set");
  for (int k = 0; k < warmup; ++k) {
    int rc = run_whole_locked(c, flags);
    if (rc) return rc;
  }
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  for (int k = 0; k < iters; ++k) {
    int rc = run_whole_locked(c, flags);
    if (rc) return rc;
  }
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(hipEventSynchronize(c->ev1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *ms_per_iter = ms / (float)iters;
  return TE_OK;
}

int te_time_chain_samples(te_ctx* c, unsigned flags, int warmup, int iters, float* ms) {
  if (!c || !ms || iters <= 0 || warmup < 0) return fail(TE_ERR_INVALID_ARG, "te_time_chain_samples: bad argument");
  CtxLock lk(c);
  if (!c->have_geo) return fail(TE_ERR_NOT_READY, "te_time_chain_samples: geometry not set");
  for (int k = 0; k < warmup; ++k) {
    int rc = run_whole_locked(c, flags);
    if (rc) return rc;
  }
  for (int k = 0; k < iters; ++k) {
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    int rc = run_whole_locked(c, flags);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(&ms[k], c->ev0, c->ev1));
  }
  return TE_OK;
}

}  // extern "C"
