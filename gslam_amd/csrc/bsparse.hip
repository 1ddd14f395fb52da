// Block-sparse Cholesky with a dense root (see bsparse.h).  gfx950 only.
//
// Numeric factorisation, one round = two launches, no atomics anywhere:
//   bs_factor_cols   one workgroup per column c of the round: every thread factorises the 7 x 7 diagonal block in registers
//                    (84 flops -- cheaper than sharing it) and thread t takes row t % 7 of slot t / 7:  L_rc = H_rc L_cc^-T
//                    (a 7-step substitution).  Lanes 0..7 carry the forward substitution along:
//                    y_c = L_cc^-1 (b_c - sum_k L_ck y_k), a GATHER over the blocks of row c (all in earlier rounds).
//   bs_update        one wave per DESTINATION block touched by the round: block(r, c') -= sum L_rc L_c'c^T over the round's
//                    columns c that hold both rows, in column order, from a list the host built with the symbolic
//                    factorisation (one entry per 7 x 7 product) -- one writer per block, a fixed order of the sums.
//                    Rounds are independent sets, so a round only writes into LATER rounds and the root.
// Then the root (dense lower triangle, right-hand side in its spare row) goes through gh_potrf_solve_dev, and the rounds
// are walked backwards:
//   bs_back          8 lanes per column: x_c = L_cc^-T (y_c - sum_r L_rc^T x_r), a gather.
// Every sum has a fixed order: the factor and the solution are bitwise reproducible from run to run.
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
  std::vector<int32_t> order, cand, picked, merged, sorted, degcnt;
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
    {  // by degree, ties in vertex order: a counting sort (the degrees of the candidates are at most tau)
      degcnt.assign((size_t)tau + 2, 0);
      for (int32_t v : cand) degcnt[adj[v].size() + 1]++;
      for (int d = 0; d <= tau; ++d) degcnt[(size_t)d + 1] += degcnt[d];
      sorted.resize(cand.size());
      for (int32_t v : cand) sorted[(size_t)degcnt[adj[v].size()]++] = v;
      cand.swap(sorted);
    }
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
      st[v] = std::move(adj[v]);  // (the picked vertices are not adjacent to each other: nobody reads adj[v] below)
      adj[v] = std::vector<int32_t>();
      alive[v] = 0;
    }
    for (int32_t v : picked) {
      const std::vector<int32_t>& sv = st[v];
      pair_products += (long long)sv.size() * (sv.size() + 1) / 2;
      for (int32_t u : sv) {  // u loses v and gains the rest of v's neighbourhood: one merge of two ascending lists
        const std::vector<int32_t>& au = adj[u];
        const size_t na = au.size(), nb = sv.size();
        merged.resize(na + nb);
        const int32_t *pa = au.data(), *pb = sv.data();
        int32_t* out = merged.data();
        size_t a = 0, b = 0, n = 0;
        while (a < na && b < nb) {  // branch-light: both cursors advance on equal elements, u and v are dropped by not counting them
          const int32_t x = pa[a], y = pb[b], w = x < y ? x : y;
          a += x <= y;
          b += y <= x;
          out[n] = w;
          n += (w != u) & (w != v);
        }
        for (; a < na; ++a) {
          out[n] = pa[a];
          n += (pa[a] != u) & (pa[a] != v);
        }
        for (; b < nb; ++b) {
          out[n] = pb[b];
          n += (pb[b] != u) & (pb[b] != v);
        }
        merged.resize(n);
        adj[u].swap(merged);
      }
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
  rowptr.assign((size_t)nf + 1, 0);
  for (int g = 0; g < n_slots; ++g) rowptr[rows[g] + 1]++;
  for (int r = 0; r < nf; ++r) rowptr[r + 1] += rowptr[r];
  rowlist.assign((size_t)std::max(n_slots, 1), 0);
  {
    std::vector<int32_t> fill(rowptr.begin(), rowptr.end() - 1);
    for (int g = 0; g < n_slots; ++g) rowlist[fill[rows[g]]++] = g;  // ascending slot = ascending column
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

__global__ __launch_bounds__(64) void bs_factor_cols_kernel(const int32_t* __restrict__ colptr, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ rowlist, const int32_t* __restrict__ slot_col,
                                                            double* __restrict__ V, size_t off_slots, double* __restrict__ Ld,
                                                            double* __restrict__ y, const double* __restrict__ b, int c0,
                                                            int32_t* __restrict__ flag) {
  const int c = c0 + blockIdx.x;
  double L[28];
  const bool ok = chol7(V + 49 * (size_t)c, L);
  if (threadIdx.x < 8) {  // forward substitution of the right-hand side (lane 7 shadows lane 6)
    const int k = threadIdx.x < 7 ? threadIdx.x : 6;
    double t = b[7 * (size_t)c + k];
    for (int e = rowptr[c]; e < rowptr[c + 1]; ++e) {  // blocks (c, earlier column), ascending column
      const int g = rowlist[e];
      const double* Bk = V + off_slots + 49 * (size_t)g + k;
      const double* yk = y + 7 * (size_t)slot_col[g];
#pragma unroll
      for (int j = 0; j < 7; ++j) t -= Bk[7 * j] * yk[j];
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const double yj = __shfl(t / L[tri(j, j)], j, 8);
      double lkj = 0;
#pragma unroll
      for (int kk = j + 1; kk < 7; ++kk)
        if (k == kk) lkj = L[tri(kk, j)];
      if (k > j) t -= lkj * yj;
      else if (k == j) t = yj;
    }
    if (threadIdx.x < 7) y[7 * (size_t)c + k] = t;
  }
  if (threadIdx.x == 0) {
    if (!ok) atomicCAS(flag, 0, c + 1);
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
      for (int i = 0; i < 7; ++i) Ld[49 * (size_t)c + 7 * j + i] = i >= j ? L[tri(i, j)] : 0.0;
  }
  const int g0 = colptr[c], m = colptr[c + 1] - g0;
  for (int i = threadIdx.x; i < 7 * m; i += 64) {
    const int s = i / 7, p = i - 7 * s;
    double* Bk = V + off_slots + 49 * (size_t)(g0 + s);
    double x[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      double v = Bk[7 * k + p];
#pragma unroll
      for (int j = 0; j < k; ++j) v -= x[j] * L[tri(k, j)];
      x[k] = v / L[tri(k, k)];
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) Bk[7 * k + p] = x[k];
  }
}

// destination d of the round: V[dst_off[d] + p + |cs| q] -= sum over its sources (g1, g) of (L_g1 L_g^T)(p, q); cs < 0 marks a
// diagonal block of the root (lower triangle only)
__global__ __launch_bounds__(256) void bs_update_kernel(const int64_t* __restrict__ dst_off, const int32_t* __restrict__ dst_cs,
                                                        const int32_t* __restrict__ src_ptr, const int2* __restrict__ src,
                                                        double* __restrict__ V, size_t off_slots, int d0, int d1) {
  const int d = d0 + blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (d >= d1 || lane >= 49) return;
  const int p = lane % 7, q = lane / 7;
  double acc = 0;
  for (int e = src_ptr[d]; e < src_ptr[d + 1]; ++e) {
    const int2 sg = src[e];
    const double* B1 = V + off_slots + 49 * (size_t)sg.x + p;
    const double* B2 = V + off_slots + 49 * (size_t)sg.y + q;
#pragma unroll
    for (int k = 0; k < 7; ++k) acc += B1[7 * k] * B2[7 * k];
  }
  int cs = dst_cs[d];
  if (cs < 0) {
    cs = -cs;
    if (p < q) return;
  }
  V[dst_off[d] + p + (size_t)cs * q] -= acc;
}

// right-hand side of the root: b_r -= sum over the blocks of row r (all in sparse columns) of L_rk y_k
__global__ __launch_bounds__(256) void bs_root_rhs_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ rowlist,
                                                          const int32_t* __restrict__ slot_col, const double* __restrict__ V,
                                                          size_t off_slots, const double* __restrict__ y, double* __restrict__ b,
                                                          int ns, int nf) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 7 * (nf - ns)) return;
  const int r = ns + i / 7, k = i % 7;
  double t = b[7 * (size_t)r + k];
  for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) {
    const int g = rowlist[e];
    const double* Bk = V + off_slots + 49 * (size_t)g + k;
    const double* yk = y + 7 * (size_t)slot_col[g];
#pragma unroll
    for (int j = 0; j < 7; ++j) t -= Bk[7 * j] * yk[j];
  }
  b[7 * (size_t)r + k] = t;
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
gh_status BsSolver::prepare_host(gh_ctx* ctx) {
  nr7 = 7 * P.nr;
  ldr = (nr7 + 1 + 15) & ~15;
  off_slots = (size_t)49 * P.ns;
  off_root = (off_slots + (size_t)49 * P.n_slots + 15) & ~(size_t)15;
  n_vals = off_root + (size_t)nr7 * ldr + 16;
  // update lists: per round, the 7 x 7 products grouped by destination block (stable: column order within a destination)
  std::vector<int64_t>& upd_off = h_upd_off;
  std::vector<int32_t>&upd_cs = h_upd_cs, &upd_ptr = h_upd_ptr, &upd_src = h_upd_src;
  upd_off.clear();
  upd_cs.clear();
  upd_src.clear();
  upd_ptr.assign(1, 0);
  upd_round.assign(1, 0);
  {
    // The products of a round ordered by destination block (column position cp, then row position rr), and within a
    // destination in generation order (source column, slot, slot).  A destination column cp is fed by the columns of the
    // round that hold a slot in ROW cp -- the row lists name them, in ascending column -- and each of them contributes its
    // slots g1 >= g as a run of ascending rr: concatenate the runs and, where more than one column feeds cp, merge them with
    // a stable sort of that handful of entries.  (A sort of all the products of a round -- 0.9 M for 20 000 keyframes --
    // took 50 ms of host time per solve; the lists are the same, element for element.)
    struct Entry {
      int32_t rr, g1, g;
    };
    std::vector<Entry> loc;
    std::vector<int32_t> rowcur(P.rowptr.begin(), P.rowptr.end() - 1);  // next unread entry of every row list
    upd_src.reserve((size_t)2 * (size_t)std::max<long long>(P.pair_products, 1));
    upd_off.reserve((size_t)std::max<long long>(P.pair_products, 1));
    upd_cs.reserve(upd_off.capacity());
    upd_ptr.reserve(upd_off.capacity() + 2);
    for (int r = 0; r < P.n_rounds; ++r) {
      const int c_hi = P.round_ptr[r + 1];
      // rows of a column are positions eliminated later, and never in the same round (an independent set): cp >= c_hi
      for (int cp = c_hi; cp < P.nf; ++cp) {
        loc.clear();
        int sources = 0;
        while (rowcur[cp] < P.rowptr[cp + 1] && P.slot_col[P.rowlist[rowcur[cp]]] < c_hi) {
          const int g = P.rowlist[rowcur[cp]++], c = P.slot_col[g];
          for (int g1 = g; g1 < P.colptr[c + 1]; ++g1) loc.push_back({P.rows[g1], g1, g});
          ++sources;
        }
        if (loc.empty()) continue;
        if (sources > 1) {
          if (loc.size() <= 48) {  // (std::stable_sort takes a temporary buffer per call: not for a dozen entries)
            for (size_t a = 1; a < loc.size(); ++a) {
              const Entry e = loc[a];
              size_t b = a;
              for (; b > 0 && loc[b - 1].rr > e.rr; --b) loc[b] = loc[b - 1];
              loc[b] = e;
            }
          } else {
            std::stable_sort(loc.begin(), loc.end(), [](const Entry& a, const Entry& b) { return a.rr < b.rr; });
          }
        }
        int cur = cp < P.ns ? P.colptr[cp] : 0;  // ascending rows: walk the slots of column cp instead of searching
        for (size_t k = 0; k < loc.size(); ++k) {
          if (k == 0 || loc[k].rr != loc[k - 1].rr) {
            if (!upd_off.empty()) upd_ptr.push_back((int32_t)(upd_src.size() / 2));
            const int rr = loc[k].rr;
            size_t o = 0;
            int cs = 7;
            if (cp < P.ns && rr != cp) {  // (block_addr() without the binary search)
              while (cur < P.colptr[cp + 1] && P.rows[cur] < rr) ++cur;
              if (cur >= P.colptr[cp + 1] || P.rows[cur] != rr)
                return gh_set_error(ctx, GH_ERR_ARG, "block-sparse solver: fill block (%d, %d) missing", rr, cp);
              o = off_slots + (size_t)49 * cur;
            } else if (!block_addr(rr, cp, &o, &cs)) {
              return gh_set_error(ctx, GH_ERR_ARG, "block-sparse solver: fill block (%d, %d) missing", rr, cp);
            }
            upd_off.push_back((int64_t)o);
            upd_cs.push_back(cp >= P.ns && rr == cp ? -cs : cs);
          }
          upd_src.push_back(loc[k].g1);
          upd_src.push_back(loc[k].g);
        }
      }
      upd_round.push_back((int32_t)upd_off.size());
    }
    upd_ptr.push_back((int32_t)(upd_src.size() / 2));
    if (upd_off.empty()) upd_ptr.assign(2, 0);
  }
  return GH_OK;
}

// the uploaded lists first and in the order of note_uploads(): they form one run of the arena and travel in one DMA
bool BsSolver::alloc_dev(GraphArena& A) {
  const size_t nf7 = (size_t)7 * P.nf;
  return A.alloc(&d_rowptr, P.rowptr.size()) && A.alloc(&d_rowlist, P.rowlist.size()) && A.alloc(&d_upd_off, h_upd_off.size()) &&
         A.alloc(&d_upd_cs, h_upd_cs.size()) && A.alloc(&d_upd_src, h_upd_src.size()) && A.alloc(&d_upd_ptr, h_upd_ptr.size()) &&
         A.alloc(&d_colptr, P.colptr.size()) && A.alloc(&d_rows, P.rows.size()) && A.alloc(&d_slot_col, P.slot_col.size()) &&
         A.alloc(&d_pos, P.pos.size()) && A.alloc(&d_flag, 2) && A.alloc(&d_Ld, (size_t)49 * std::max(P.ns, 1)) && A.alloc(&d_y, nf7) &&
         A.alloc(&d_b, nf7 + 16) && A.alloc(&d_H, n_vals) && A.alloc(&d_W, n_vals);
}

void BsSolver::note_uploads(GraphArena& A) {
  A.upload(d_rowptr, P.rowptr.data(), P.rowptr.size() * 4);
  A.upload(d_rowlist, P.rowlist.data(), P.rowlist.size() * 4);
  A.upload(d_upd_off, h_upd_off.data(), h_upd_off.size() * 8);
  A.upload(d_upd_cs, h_upd_cs.data(), h_upd_cs.size() * 4);
  A.upload(d_upd_src, h_upd_src.data(), h_upd_src.size() * 4);
  A.upload(d_upd_ptr, h_upd_ptr.data(), h_upd_ptr.size() * 4);
  A.upload(d_colptr, P.colptr.data(), P.colptr.size() * 4);
  A.upload(d_rows, P.rows.data(), P.rows.size() * 4);
  A.upload(d_slot_col, P.slot_col.data(), P.slot_col.size() * 4);
  A.upload(d_pos, P.pos.data(), P.pos.size() * 4);
}

gh_status BsSolver::clear_values(gh_ctx* ctx) {
  GH_HIP(ctx, hipMemsetAsync(d_H, 0, n_vals * 8, ctx->stream));
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
    const int c0 = P.round_ptr[r], c1 = P.round_ptr[r + 1], u0 = upd_round[r], u1 = upd_round[r + 1];
    GH_LAUNCH(ctx, "bs_factor_cols", bs_factor_cols_kernel, dim3(c1 - c0), dim3(64), 0, (const int32_t*)d_colptr, (const int32_t*)d_rowptr,
              (const int32_t*)d_rowlist, (const int32_t*)d_slot_col, d_W, off_slots, d_Ld, d_y, (const double*)d_b, c0, d_flag);
    if (u1 > u0)
      GH_LAUNCH(ctx, "bs_update", bs_update_kernel, dim3(gh_div_up(u1 - u0, 4)), dim3(256), 0, (const int64_t*)d_upd_off,
                (const int32_t*)d_upd_cs, (const int32_t*)d_upd_ptr, (const int2*)d_upd_src, d_W, off_slots, u0, u1);
  }
  if (P.nr > 0) {
    int dinfo = 0;
    if (P.ns > 0)
      GH_LAUNCH(ctx, "bs_root_rhs", bs_root_rhs_kernel, dim3(gh_div_up(nr7, 256)), dim3(256), 0, (const int32_t*)d_rowptr,
                (const int32_t*)d_rowlist, (const int32_t*)d_slot_col, (const double*)d_W, off_slots, (const double*)d_y, d_b, P.ns, P.nf);
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
  int32_t* hf = h_flag ? h_flag : &flag;  // (pinned when the caller has a read-back block: a plain DMA, no staging)
  GH_HIP(ctx, hipMemcpyAsync(hf, d_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *info = *hf;
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
  GH_TRY(S.prepare_host(ctx));
  GraphArena A(ctx);
  double *d_g = nullptr, *d_x = nullptr;
  auto alloc_all = [&]() { return S.alloc_dev(A) && A.alloc(&d_g, (size_t)7 * n_frames) && A.alloc(&d_x, (size_t)7 * n_frames); };
  alloc_all();  // measuring pass
  GH_TRY(A.reserve());
  if (!alloc_all())
    return gh_set_error(ctx, GH_ERR_NOMEM, "block-sparse solver: device allocation failed (%d sparse columns, %d slots, root %d)",
                        S.P.ns, S.P.n_slots, S.nr7);
  S.note_uploads(A);
  GH_TRY(A.flush());
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // (the staging block is free again)
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
  GH_HIP(ctx, hipMemcpyAsync(S.d_H, vals.data(), S.n_vals * 8, hipMemcpyHostToDevice, ctx->stream));
  GH_HIP(ctx, hipMemcpyAsync(d_g, g, (size_t)7 * n_frames * 8, hipMemcpyHostToDevice, ctx->stream));
  GH_TRY(S.factor_solve(ctx, radius, d_g, d_x, info));
  GH_HIP(ctx, hipMemcpyAsync(x_out, d_x, (size_t)7 * n_frames * 8, hipMemcpyDeviceToHost, ctx->stream));
  GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GH_OK;
}
