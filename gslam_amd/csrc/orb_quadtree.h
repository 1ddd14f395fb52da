// Internal interface between orb.hip and orb_quadtree.hip (the ORB-SLAM style keypoint distribution,
// gh_orb_plan_set_distribution(plan, 1); specification: oracle/orb_oracle.c steps 4' and 5').
#pragma once
#include "common.h"
#include "gslam_orb_tables.h"

struct LevelView {
  const uint8_t* base;   // frame 0
  size_t frame_stride;   // bytes between frames
  int pitch, w, h;
};

// one selected keypoint of a level, in level coordinates (select / quadtree -> describe)
struct SelKp {
  uint16_t x, y;
  uint8_t score, level;
  uint16_t pad;
};

struct gh_qt_plan;

// Score planes (orb.hip's tile kernel -> slam_cells_plane_kernel): pixel (y, x) of a level lives at base[y * pitch + x + kQtPlaneX].
// A 64 x 64 tile starts at x = GH_ORB_EDGE + 64 bx, so with 45 = 64 - 19 a tile row is one 64-byte aligned segment of the plane.
constexpr int kQtPlaneX = 64 - GH_ORB_EDGE;
static_assert(kQtPlaneX % 4 == 1, "the tile kernel's dword stores and the cell kernel's dword loads assume x + kQtPlaneX = x + 1 (mod 4)");

// Buffers of the quadtree mode for `max_batch` frames of the given pyramid.  *bytes += device bytes allocated.
gh_status gh_qt_create(gh_ctx* ctx, int n_levels, const int* lw, const int* lh, const int* quota, int max_batch,
                       gh_qt_plan** out, size_t* bytes);
void gh_qt_destroy(gh_qt_plan* q);
// Steps 4' and 5' for levels 0 .. n_levels-1 of `batch` frames on ctx->stream: sel[b * K + quota_off[l] + i], level_cnt[b * 8 + l]
// exactly as orb_select leaves them.
// planes (may be null): per level the score plane orb.hip's tile kernel wrote (base = null: none for that level; pixel (y, x)
// at base[y * pitch + x + kQtPlaneX], S of oracle step 2 / 3) -- the cells of such a level are taken from it instead of the image.
gh_status gh_qt_enqueue(gh_ctx* ctx, gh_qt_plan* q, const LevelView* lv, int batch, int min_th, int ini_th,
                        const int* quota_off, int K, SelKp* sel, int32_t* level_cnt, const LevelView* planes = nullptr);
bool gh_qt_can_overflow(const gh_qt_plan* q);
bool gh_qt_plane_ok(const gh_qt_plan* q, int l);
// gh_qt_enqueue in three parts (each on ctx->stream at the time of the call): counters, the cells of ONE level (plane may be
// null / hold a null base), the tree -- for a caller that runs the cells of level l beside the kernel producing level l + 1
gh_status gh_qt_begin(gh_ctx* ctx, gh_qt_plan* q, int batch);
gh_status gh_qt_cells(gh_ctx* ctx, gh_qt_plan* q, int l, const LevelView& img, const LevelView* plane, int batch, int min_th, int ini_th);
gh_status gh_qt_tree(gh_ctx* ctx, gh_qt_plan* q, int batch, const int* quota_off, int K, SelKp* sel, int32_t* level_cnt);
// After the stream has drained: GH_ERR_RANGE-style failure if a candidate list overflowed its buffer (never silent).
gh_status gh_qt_check(gh_ctx* ctx, gh_qt_plan* q);
