// Where element (r, c) of the reduced camera system lives, for the band / arrowhead solver (chol_cr.hip) and for the kernels
// of ba.hip that assemble the system.  Two layouts behind one addressing rule, element (r, c) at  A[c * lda + r + shift]:
//
//   DENSE   (m == 0)  the column-major lower triangle of chol.hip: shift = 0, lda > n.  The test / tool entry points
//                     (gh_band_solve_dev, gh_arrow_solve_dev) and rounds 4-5 of gh_ba_solve: n x lda doubles -- 28.8 GB at C5,
//                     115 GB at 20 k cameras -- of which block cyclic reduction ever touches a band.
//   COMPACT (m > 0)   every column keeps only what the reduction touches, in `lda` = (2 + levels) m + nbr + 1 rows (padded):
//                       local rows [0, 2 m)                the rows of the column's own superblock J = c / m and of J + 1
//                                                          (the band: half-width <= m), local row = r - J m
//                       local rows [(1 + k) m, (2 + k) m)  k = 1 .. levels: the fill block B(J + 2^k, J) that eliminating J + 2^(k-1)
//                                                          at level 2^(k-1) leaves (only columns with J % 2^k == 0 use the slot)
//                       local rows [brow, brow + nbr]      the border rows (arrowhead systems) and the right-hand side row
//                     0.93 GB at C5, 2.0 GB at 20 k cameras.  Border COLUMNS (c >= n_band) use the same rule: only their border
//                     rows exist.
// A kernel works on one (row superblock I, column superblock J) pair at a time, so the shift is a per-block constant: it is
// added to the base pointer and the kernel's index arithmetic in GLOBAL rows and columns stays what it was.
#pragma once

struct CrMap {
  int m = 0;       // superblock columns (64 T); 0 = dense layout
  int n_band = 0;  // band unknowns: the border rows / the right-hand-side row start at this global row
  int brow = 0;    // compact: local row of global row n_band
  int levels = 0;  // compact: fill slots = reduction levels (log2 of the survivors' stride: the dense top takes the rest)
#if defined(__HIPCC__) || defined(__CUDACC__)
  __host__ __device__
#endif
  long long shift(int I, int J) const {  // rows of superblock I in a column of superblock J (I - J is 0, 1 or a power of two)
    if (m == 0) return 0;
    const int d = I - J;
    if (d <= 1) return -(long long)J * m;
    int k = 0;
    while ((2 << k) <= d) ++k;  // d = 2^k
    return (long long)(1 + k) * m - (long long)I * m;
  }
#if defined(__HIPCC__) || defined(__CUDACC__)
  __host__ __device__
#endif
  long long bshift() const { return m == 0 ? 0 : (long long)brow - n_band; }  // border rows and the right-hand-side row
};

// rows a compact column needs: the band, `levels` fill slots, nbr border rows and the right-hand side, padded to 128-byte lines
inline int cr_compact_lda(int m, int levels, int nbr) { return ((2 + levels) * m + nbr + 1 + 15) & ~15; }
