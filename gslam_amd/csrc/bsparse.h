// Block-sparse Cholesky (7 x 7 keyframe blocks) with a dense root -- the linear solver of LARGE pose graphs
// (GSLAM/core/Optimizer.h:127-148,162-167: se3Graph / sim3Graph / gpsGraph over thousands of keyframes).  Internal to
// libgslam_hip.so; the public handle on it is gh_bs_symbolic / gh_bs_solve_host (tests) and gh_pg_solve.
//
// Ordering: rounds of "multiple elimination".  Every round eliminates an INDEPENDENT set of low-degree keyframes (no two
// adjacent in the current elimination graph), so all columns of a round factorise in parallel -- one grid per round
// instead of one dependent step per keyframe; on an odometry chain this is cyclic reduction (log2 rounds).  The rounds
// stop when what is left is small or dense; that remainder (the root) is copied nowhere: its Schur complement is built in
// place in a dense lower-triangular matrix and goes through the blocked MFMA Cholesky of chol.hip.
#pragma once
#include "common.h"
#include "graph_arena.h"

struct BsPattern {  // host-side symbolic factorisation
  int nf = 0, ns = 0, nr = 0, n_rounds = 0, n_slots = 0;
  std::vector<int32_t> pos;        // frame -> position in the elimination order (root frames last)
  std::vector<int32_t> round_ptr;  // n_rounds + 1: positions [round_ptr[r], round_ptr[r + 1]) are eliminated in round r
  std::vector<int32_t> colptr;     // ns + 1: the below-diagonal blocks (slots) of sparse column c are colptr[c] .. colptr[c + 1]
  std::vector<int32_t> rows;       // n_slots: row POSITION of each slot, ascending within a column
  std::vector<int32_t> slot_col;   // n_slots: the column a slot belongs to
  std::vector<int32_t> rowptr, rowlist;  // by row position (sparse and root): the slots in that row, ascending column
  long long pair_products = 0;     // 7 x 7 block products of the numeric factorisation (work estimate)
  void build(int n_frames, int n_pairs, const int32_t* prow, const int32_t* pcol, int root_min, int max_rounds);
  // slot of row position r in sparse column c (-1: structurally zero)
  int find(int c, int r) const;
};

// Values live in ONE device buffer of doubles: [diagonal blocks ns x 49 | slot blocks n_slots x 49 | root nr7 x ldr],
// every 7 x 7 block column-major (element (row a, column b) at 7 b + a), the root column-major lower with leading
// dimension ldr (one spare row: the right-hand side rides through the dense factorisation).
struct BsSolver {
  BsPattern P;
  int nr7 = 0, ldr = 0;
  size_t off_slots = 0, off_root = 0, n_vals = 0;
  int32_t *d_colptr = nullptr, *d_rows = nullptr, *d_slot_col = nullptr, *d_pos = nullptr, *d_flag = nullptr;
  int32_t *d_rowptr = nullptr, *d_rowlist = nullptr;
  // update lists: destinations of round r are upd_round[r] .. upd_round[r + 1]; destination d takes the products of the
  // slot pairs d_upd_src[d_upd_ptr[d] .. d_upd_ptr[d + 1]) in that order
  std::vector<int32_t> upd_round;
  int64_t* d_upd_off = nullptr;
  int32_t *d_upd_cs = nullptr, *d_upd_ptr = nullptr, *d_upd_src = nullptr;
  double *d_H = nullptr, *d_W = nullptr;  // assembled values / damped working copy that becomes the factor
  double *d_Ld = nullptr, *d_y = nullptr, *d_b = nullptr;
  // host images of the update lists (kept until the uploads noted by note_uploads() have been flushed)
  std::vector<int64_t> h_upd_off;
  std::vector<int32_t> h_upd_cs, h_upd_ptr, h_upd_src;
  int32_t* h_flag = nullptr;  // pinned word the pivot flag of a factorisation is read back into (nullptr: a local)
  // After P.build, in this order: prepare_host (sizes and update lists), alloc_dev (once with the arena measuring, once for
  // real), note_uploads + A.flush() + clear_values; the arena owns the device memory.
  gh_status prepare_host(gh_ctx* ctx);
  bool alloc_dev(GraphArena& A);
  void note_uploads(GraphArena& A);
  gh_status clear_values(gh_ctx* ctx);
  // offset (in doubles from d_H / d_W) and strides of the block (row frame position pr, column frame position pc), pr >= pc:
  // element (a, b) of the block is at off + a + cs * b
  bool block_addr(int pr, int pc, size_t* off, int* cs) const;
  // W = H + clamp(diag) / radius; solves W x = -g (g, x indexed by FRAME: 7 f + k).  *info: 0 ok, else 1-based position of a
  // non-positive pivot (sparse part) or nf + dense info (root)
  gh_status factor_solve(gh_ctx* ctx, double radius, const double* g_dev, double* x_dev, int* info);
};
