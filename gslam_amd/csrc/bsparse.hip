// Block-sparse Cholesky with a dense root (see bsparse.h).  gfx950 only.
//
// Numeric factorisation, one round = two launches:
//   bs_factor_cols   one workgroup per column c of the round: every thread factorises the 7 x 7 diagonal block in registers
//                    (84 flops -- cheaper than sharing it), thread t takes row t % 7 of slot t / 7:  L_rc = H_rc L_cc^-T
//                    (a 7-step substitution), and carries the forward substitution of the right-hand side along:
//                    y_c = L_cc^-1 b_c,  b_r -= L_rc y_c  (f64 atomics: columns of one round share rows)
//   bs_update        one workgroup per slot (c, c'): for every slot (c, r) below it, block(r, c') -= L_rc L_c'c^T, 49 lanes
//                    one element each, f64 atomics into the destination column's slot (binary search in its sorted row
//                    list), its diagonal block, or the dense root.  Rounds are independent sets, so a round only writes
//                    into LATER rounds and the root.
// Then the root (dense lower triangle, right-hand side in its spare row) goes through gh_potrf_solve_dev, and the rounds
// are walked backwards:
//   bs_back          8 lanes per column: x_c = L_cc^-T (y_c - sum_r L_rc^T x_r), a gather (no atomics).
// The sums that meet in one block come from different columns in scheduling order: the factor is reproducible to
// rounding, not bit for bit (the dense path, used below GSLAM_HIP_PG_SPARSE_MIN keyframes, is).
#include "bsparse.h"

#include <algorithm>
#include <climits>
#include <iterator>

// ---------------------------------------------------------------- symbolic factorisation (host)
void BsPattern::build(int n_frames, int n_pairs, const int32_t* prow, const int32_t* pcol, int root_min, int max_rounds) {
  nf = n_frames;
  std::vector<std::vector<int32_t>> adj((size_t)nf), st((size_t)nf);
  for (int k = 0; k < n_pairs; ++k) {
    const int a = prow[k], b = pcol[k];
    if (a == b || a < 0 || b < 0 || a >= nf || b >= nf) continue;
    adj[a].push_back(b);
    adj[b].push_back(a);
  }
  for (auto& v : adj) {
    std::sort(v.begin(), v.end());
    v.erase(std::unique(v.begin(), v.end()), v.end());
  }
  std::vector<char> alive((size_t)nf, 1), blocked((size_t)nf, 0);
  std::vector<int32_t> order, cand, picked, merged;
  order.reserve((size_t)nf);
  round_ptr.assign(1, 0);
  int remaining = nf;
  if (root_min < 1) root_min = 1;
  while (remaining > root_min && (int)round_ptr.size() <= max_rounds) {
    int dmin = INT_MAX;
    for (int v = 0; v < nf; ++v)
      if (alive[v]) dmin = std::min(dmin, (int)adj[v].size());
    if ((long long)dmin * 3 > remaining) break;  // what is left is dense: it is the root
    const int tau = dmin + std::max(2, dmin / 2);
    cand.clear();
    for (int v = 0; v < nf; ++v)
      if (alive[v] && (int)adj[v].size() <= tau && (long long)adj[v].size() * 3 <= remaining) cand.push_back(v);
    std::stable_sort(cand.begin(), cand.end(), [&](int32_t a, int32_t b) { return adj[a].size() < adj[b].size(); });
    std::fill(blocked.begin(), blocked.end(), 0);
    picked.clear();
    const int room = remaining - root_min;  // never eat into the root's minimum
    for (int32_t v : cand) {
      if (blocked[v] || (int)picked.size() >= room) continue;
      picked.push_back(v);
      blocked[v] = 1;
      for (int32_t u : adj[v]) blocked[u] = 1;
    }
    if (picked.empty()) break;
    std::sort(picked.begin(), picked.end());
    for (int32_t v : picked) {
      st[v] = adj[v];
      alive[v] = 0;
    }
    for (int32_t v : picked) {
      const std::vector<int32_t>& sv = st[v];
      pair_products += (long long)sv.size() * (sv.size() + 1) / 2;
      for (int32_t u : sv) {  // u loses v and gains the rest of v's neighbourhood
        merged.clear();
        std::set_union(adj[u].begin(), adj[u].end(), sv.begin(), sv.end(), std::back_inserter(merged));
        merged.erase(std::remove_if(merged.begin(), merged.end(), [&](int32_t w) { return w == u || w == v; }), merged.end());
        adj[u].swap(merged);
      }
      adj[v].clear();
      adj[v].shrink_to_fit();
    }
    remaining -= (int)picked.size();
    order.insert(order.end(), picked.begin(), picked.end());
    round_ptr.push_back((int32_t)order.size());
  }
  ns = (int)order.size();
  nr = nf - ns;
  n_rounds = (int)round_ptr.size() - 1;
  pos.assign((size_t)nf, 0);
  for (int c = 0; c < ns; ++c) pos[order[c]] = c;
  {
    int p = ns;
    for (int v = 0; v < nf; ++v)
      if (alive[v]) pos[v] = p++;
  }
  colptr.assign((size_t)ns + 1, 0);
  for (int c = 0; c < ns; ++c) colptr[c + 1] = colptr[c] + (int32_t)st[order[c]].size();
  n_slots = colptr[ns];
  rows.assign((size_t)std::max(n_slots, 1), 0);
  slot_col.assign((size_t)std::max(n_slots, 1), 0);
  for (int c = 0; c < ns; ++c) {
    int32_t* r = rows.data() + colptr[c];
    const std::vector<int32_t>& sv = st[order[c]];
    for (size_t k = 0; k < sv.size(); ++k) {
      r[k] = pos[sv[k]];
      slot_col[colptr[c] + k] = c;
    }
    std::sort(r, r + sv.size());
  }
}

int BsPattern::find(int c, int r) const {
  const int32_t* b = rows.data() + colptr[c];
  const int32_t* e = rows.data() + colptr[c + 1];
  const int32_t* it = std::lower_bound(b, e, (int32_t)r);
  return (it != e && *it == r) ? (int)(it - rows.data()) : -1;
}

// ---------------------------------------------------------------- kernels
namespace {

__device__ __host__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // packed lower triangle, j <= i

// L L^T = D (lower triangle of the column-major 7 x 7 block read); a non-positive pivot is replaced by 1 and reported
__device__ inline bool chol7(const double* __restrict__ D, double* L) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    double d = D[7 * j + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[tri(j, k)] * L[tri(j, k)];
    if (!(d > 0.0) || !(d < 1e300)) {
      ok = false;
      d = 1.0;
    }
    const double s = sqrt(d);
    L[tri(j, j)] = s;
#pragma unroll
    for (int i = j + 1; i < 7; ++i) {
      double v = D[7 * j + i];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= L[tri(i, k)] * L[tri(j, k)];
      L[tri(i, j)] = v / s;
    }
  }
  return ok;
}

__global__ __launch_bounds__(64) void bs_factor_cols_kernel(const int32_t* __restrict__ colptr, const int32_t* __restrict__ rows,
                                                            double* __restrict__ V, size_t off_slots, double* __restrict__ Ld,
                                                            double* __restrict__ y, double* __restrict__ b, int c0,
                                                            int32_t* __restrict__ flag) {
  const int c = c0 + blockIdx.x;
  double L[28], yc[7];
  const bool ok = chol7(V + 49 * (size_t)c, L);
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    double v = b[7 * (size_t)c + k];
#pragma unroll
    for (int j = 0; j < k; ++j) v -= L[tri(k, j)] * yc[j];
    yc[k] = v / L[tri(k, k)];
  }
  if (threadIdx.x == 0) {
    if (!ok) atomicCAS(flag, 0, c + 1);
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
      for (int i = 0; i < 7; ++i) Ld[49 * (size_t)c + 7 * j + i] = i >= j ? L[tri(i, j)] : 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) y[7 * (size_t)c + k] = yc[k];
  }
  const int g0 = colptr[c], m = colptr[c + 1] - g0;
  for (int i = threadIdx.x; i < 7 * m; i += 64) {
    const int s = i / 7, p = i - 7 * s;
    double* Bk = V + off_slots + 49 * (size_t)(g0 + s);
    double x[7], dot = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      double v = Bk[7 * k + p];
#pragma unroll
      for (int j = 0; j < k; ++j) v -= x[j] * L[tri(k, j)];
      x[k] = v / L[tri(k, k)];
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      Bk[7 * k + p] = x[k];
      dot += x[k] * yc[k];
    }
    atomicAdd(&b[7 * (size_t)rows[g0 + s] + p], -dot);
  }
}

__global__ __launch_bounds__(256) void bs_update_kernel(const int32_t* __restrict__ colptr, const int32_t* __restrict__ rows,
                                                        const int32_t* __restrict__ slot_col, double* __restrict__ V,
                                                        size_t off_slots, size_t off_root, int ns, int ldr, int g_first) {
  const int g = g_first + blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane >= 49) return;  // (no barrier below)
  const int c = slot_col[g], gc1 = colptr[c + 1];
  const int cp = rows[g];  // destination column (a position)
  const int p = lane % 7, q = lane / 7;
  double b2[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) b2[k] = V[off_slots + 49 * (size_t)g + 7 * k + q];
  int d0 = 0, d1 = 0;
  if (cp < ns) {
    d0 = colptr[cp];
    d1 = colptr[cp + 1];
  }
  for (int g1 = g + wave; g1 < gc1; g1 += 4) {
    const int r = rows[g1];
    const double* B1 = V + off_slots + 49 * (size_t)g1 + p;
    double v = 0;
#pragma unroll
    for (int k = 0; k < 7; ++k) v += B1[7 * k] * b2[k];
    double* dst;
    if (cp >= ns) {
      if (r == cp && p < q) continue;  // the root keeps its lower triangle only
      dst = V + off_root + (size_t)(7 * (cp - ns) + q) * ldr + 7 * (r - ns) + p;
    } else if (r == cp) {
      dst = V + 49 * (size_t)cp + 7 * q + p;
    } else {
      int lo = d0, hi = d1;  // first slot of column cp whose row is >= r: it IS r (fill property)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (rows[mid] < r) lo = mid + 1;
        else hi = mid;
      }
      dst = V + off_slots + 49 * (size_t)lo + 7 * q + p;
    }
    atomicAdd(dst, -v);
  }
}

__global__ __launch_bounds__(64) void bs_back_kernel(const int32_t* __restrict__ colptr, const int32_t* __restrict__ rows,
                                                     const double* __restrict__ V, size_t off_slots, const double* __restrict__ Ld,
                                                     const double* __restrict__ y, double* __restrict__ x, int c0, int c1) {
  const int cl = c0 + blockIdx.x * 8 + (threadIdx.x >> 3), kl = threadIdx.x & 7;
  const bool live = cl < c1 && kl < 7;
  const int c = cl < c1 ? cl : c1 - 1, k = kl < 7 ? kl : 6;  // idle lanes shadow a live one: the shuffles need every lane
  double t = y[7 * (size_t)c + k];
  for (int g = colptr[c]; g < colptr[c + 1]; ++g) {
    const double* Bk = V + off_slots + 49 * (size_t)g + 7 * k;
    const double* xr = x + 7 * (size_t)rows[g];
#pragma unroll
    for (int p = 0; p < 7; ++p) t -= Bk[p] * xr[p];
  }
  const double* L = Ld + 49 * (size_t)c;
#pragma unroll
  for (int j = 6; j >= 0; --j) {
    const double xj = __shfl(t / L[7 * j + j], j, 8);
    if (k < j) t -= L[7 * k + j] * xj;
    else if (k == j) t = xj;
  }
  if (live) x[7 * (size_t)c + k] = t;
}

// damping of the working copy + right-hand side in elimination order
__global__ __launch_bounds__(256) void bs_prepare_kernel(const int32_t* __restrict__ pos, int nf, int ns, double* __restrict__ W,
                                                         size_t off_root, int ldr, double radius, const double* __restrict__ g,
                                                         double* __restrict__ b) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 7 * nf) return;
  const int f = i / 7, k = i - 7 * f, c = pos[f];
  double* d = c < ns ? W + 49 * (size_t)c + 8 * k : W + off_root + (size_t)(7 * (c - ns) + k) * ldr + 7 * (c - ns) + k;
  const double v = *d;
  *d = v + (v < 1e-6 ? 1e-6 : (v > 1e32 ? 1e32 : v)) / radius;
  b[7 * (size_t)c + k] = -g[i];
}

__global__ __launch_bounds__(256) void bs_unpermute_kernel(const int32_t* __restrict__ pos, int nf, const double* __restrict__ b,
                                                           double* __restrict__ x) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 7 * nf) return;
  const int f = i / 7, k = i - 7 * f;
  x[i] = b[7 * (size_t)pos[f] + k];
}

}  // namespace

// ---------------------------------------------------------------- solver object
BsSolver::~BsSolver() {
  for (void* p : owned) (void)hipFree(p);
}

gh_status BsSolver::init(gh_ctx* ctx) {
  nr7 = 7 * P.nr;
  ldr = (nr7 + 1 + 15) & ~15;
  off_slots = (size_t)49 * P.ns;
  off_root = (off_slots + (size_t)49 * P.n_slots + 15) & ~(size_t)15;
  n_vals = off_root + (size_t)nr7 * ldr + 16;
  auto alloc = [&](void** out, size_t bytes) -> bool {
    if (hipMalloc(out, bytes ? bytes : 8) != hipSuccess) return false;
    owned.push_back(*out);
    return true;
  };
  const size_t nf7 = (size_t)7 * P.nf;
  const bool ok = alloc((void**)&d_colptr, P.colptr.size() * 4) && alloc((void**)&d_rows, P.rows.size() * 4) &&
                  alloc((void**)&d_slot_col, P.slot_col.size() * 4) && alloc((void**)&d_pos, P.pos.size() * 4) &&
                  alloc((void**)&d_flag, 8) && alloc((void**)&d_H, n_vals * 8) && alloc((void**)&d_W, n_vals * 8) &&
                  alloc((void**)&d_Ld, (size_t)49 * std::max(P.ns, 1) * 8) && alloc((void**)&d_y, nf7 * 8) && alloc((void**)&d_b, (nf7 + 16) * 8);
  if (!ok)
    return gh_set_error(ctx, GH_ERR_NOMEM, "block-sparse solver: device allocation failed (%d sparse columns, %d slots, root %d)", P.ns,
                        P.n_slots, nr7);
  GH_HIP(ctx, hipMemcpyAsync(d_colptr, P.colptr.data(), P.colptr.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  GH_HIP(ctx, hipMemcpyAsync(d_rows, P.rows.data(), P.rows.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  GH_HIP(ctx, hipMemcpyAsync(d_slot_col, P.slot_col.data(), P.slot_col.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  GH_HIP(ctx, hipMemcpyAsync(d_pos, P.pos.data(), P.pos.size() * 4, hipMemcpyHostToDevice, ctx->stream));
  GH_HIP(ctx, hipMemsetAsync(d_H, 0, n_vals * 8, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GH_OK;
}

bool BsSolver::block_addr(int pr, int pc, size_t* off, int* cs) const {
  if (pr < pc) return false;
  if (pc >= P.ns) {
    *off = off_root + (size_t)(7 * (pc - P.ns)) * ldr + 7 * (pr - P.ns);
    *cs = ldr;
    return true;
  }
  *cs = 7;
  if (pr == pc) {
    *off = (size_t)49 * pc;
    return true;
  }
  const int slot = P.find(pc, pr);
  if (slot < 0) return false;
  *off = off_slots + (size_t)49 * slot;
  return true;
}

gh_status BsSolver::factor_solve(gh_ctx* ctx, double radius, const double* g_dev, double* x_dev, int* info) {
  *info = 0;
  GH_HIP(ctx, hipMemcpyAsync(d_W, d_H, n_vals * 8, hipMemcpyDeviceToDevice, ctx->stream));
  GH_HIP(ctx, hipMemsetAsync(d_flag, 0, 8, ctx->stream));
  GH_LAUNCH(ctx, "bs_prepare", bs_prepare_kernel, dim3(gh_div_up(7 * P.nf, 256)), dim3(256), 0, (const int32_t*)d_pos, P.nf, P.ns, d_W,
            off_root, ldr, radius, g_dev, d_b);
  for (int r = 0; r < P.n_rounds; ++r) {
    const int c0 = P.round_ptr[r], c1 = P.round_ptr[r + 1], g0 = P.colptr[c0], g1 = P.colptr[c1];
    GH_LAUNCH(ctx, "bs_factor_cols", bs_factor_cols_kernel, dim3(c1 - c0), dim3(64), 0, (const int32_t*)d_colptr, (const int32_t*)d_rows,
              d_W, off_slots, d_Ld, d_y, d_b, c0, d_flag);
    if (g1 > g0)
      GH_LAUNCH(ctx, "bs_update", bs_update_kernel, dim3(g1 - g0), dim3(256), 0, (const int32_t*)d_colptr, (const int32_t*)d_rows,
                (const int32_t*)d_slot_col, d_W, off_slots, off_root, P.ns, ldr, g0);
  }
  if (P.nr > 0) {
    int dinfo = 0;
    GH_TRY(gh_potrf_solve_dev(ctx, d_W + off_root, nr7, ldr, d_b + (size_t)7 * P.ns, &dinfo));
    if (dinfo) {
      *info = P.nf + dinfo;
      return GH_OK;
    }
  }
  for (int r = P.n_rounds - 1; r >= 0; --r) {
    const int c0 = P.round_ptr[r], c1 = P.round_ptr[r + 1];
    GH_LAUNCH(ctx, "bs_back", bs_back_kernel, dim3(gh_div_up(c1 - c0, 8)), dim3(64), 0, (const int32_t*)d_colptr, (const int32_t*)d_rows,
              (const double*)d_W, off_slots, (const double*)d_Ld, (const double*)d_y, d_b, c0, c1);
  }
  GH_LAUNCH(ctx, "bs_unpermute", bs_unpermute_kernel, dim3(gh_div_up(7 * P.nf, 256)), dim3(256), 0, (const int32_t*)d_pos, P.nf,
            (const double*)d_b, x_dev);
  int32_t flag = 0;
  GH_HIP(ctx, hipMemcpyAsync(&flag, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *info = flag;
  return GH_OK;
}

// ---------------------------------------------------------------- C ABI (tests, tools)
extern "C" gh_status gh_bs_symbolic(int n_frames, int n_pairs, const int32_t* prow, const int32_t* pcol, int root_min, int max_rounds,
                                    int32_t* pos_out, int64_t* counts_out, int32_t* round_ptr_out, int round_cap, int32_t* colptr_out,
                                    int32_t* rows_out, int rows_cap) {
  if (n_frames < 1 || n_pairs < 0 || (n_pairs && (!prow || !pcol)) || !counts_out) return GH_ERR_ARG;
  BsPattern P;
  P.build(n_frames, n_pairs, prow, pcol, root_min, max_rounds);
  counts_out[0] = P.ns;
  counts_out[1] = P.nr;
  counts_out[2] = P.n_rounds;
  counts_out[3] = P.n_slots;
  counts_out[4] = P.pair_products;
  if (pos_out) memcpy(pos_out, P.pos.data(), (size_t)n_frames * 4);
  if (round_ptr_out) {
    if (round_cap < P.n_rounds + 1) return GH_ERR_ARG;
    memcpy(round_ptr_out, P.round_ptr.data(), (size_t)(P.n_rounds + 1) * 4);
  }
  if (colptr_out) memcpy(colptr_out, P.colptr.data(), (size_t)(P.ns + 1) * 4);
  if (rows_out) {
    if (rows_cap < P.n_slots) return GH_ERR_ARG;
    memcpy(rows_out, P.rows.data(), (size_t)P.n_slots * 4);
  }
  return GH_OK;
}

extern "C" gh_status gh_bs_solve_host(gh_ctx* ctx, int n_frames, int n_pairs, const int32_t* prow, const int32_t* pcol,
                                      const double* diag, const double* off, const double* g, double radius, int root_min,
                                      int max_rounds, double* x_out, int* info) {
  if (!ctx) return GH_ERR_ARG;
  GH_ENTER(ctx);
  GH_CHECK_ARG(ctx, n_frames >= 1 && n_pairs >= 0 && diag && g && x_out && info && radius > 0 && (n_pairs == 0 || (prow && pcol && off)));
  BsSolver S;
  S.P.build(n_frames, n_pairs, prow, pcol, root_min, max_rounds);
  GH_TRY(S.init(ctx));
  std::vector<double> vals(S.n_vals, 0.0);
  for (int f = 0; f < n_frames; ++f) {
    size_t o;
    int cs;
    GH_CHECK_ARG(ctx, S.block_addr(S.P.pos[f], S.P.pos[f], &o, &cs));
    for (int b = 0; b < 7; ++b)
      for (int a = 0; a < 7; ++a)
        if (cs == 7 || a >= b) vals[o + a + (size_t)cs * b] = diag[49 * (size_t)f + 7 * b + a];
  }
  for (int k = 0; k < n_pairs; ++k) {  // block (row frame prow[k], column frame pcol[k]), column-major
    const int pa = S.P.pos[prow[k]], pb = S.P.pos[pcol[k]];
    GH_CHECK_ARG(ctx, pa != pb);
    size_t o;
    int cs;
    GH_CHECK_ARG(ctx, S.block_addr(std::max(pa, pb), std::min(pa, pb), &o, &cs));
    for (int b = 0; b < 7; ++b)
      for (int a = 0; a < 7; ++a) {
        const double v = off[49 * (size_t)k + 7 * b + a];
        vals[pa > pb ? o + a + (size_t)cs * b : o + b + (size_t)cs * a] += v;
      }
  }
  double *d_g = nullptr, *d_x = nullptr;
  GH_HIP(ctx, hipMalloc((void**)&d_g, (size_t)7 * n_frames * 8));
  S.owned.push_back(d_g);
  GH_HIP(ctx, hipMalloc((void**)&d_x, (size_t)7 * n_frames * 8));
  S.owned.push_back(d_x);
  GH_HIP(ctx, hipMemcpyAsync(S.d_H, vals.data(), S.n_vals * 8, hipMemcpyHostToDevice, ctx->stream));
  GH_HIP(ctx, hipMemcpyAsync(d_g, g, (size_t)7 * n_frames * 8, hipMemcpyHostToDevice, ctx->stream));
  GH_TRY(S.factor_solve(ctx, radius, d_g, d_x, info));
  GH_HIP(ctx, hipMemcpyAsync(x_out, d_x, (size_t)7 * n_frames * 8, hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GH_OK;
}
