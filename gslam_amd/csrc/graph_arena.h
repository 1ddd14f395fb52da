// Device buffers and host -> device uploads of ONE graph solve (gh_graph_solve / gh_pg_solve / the block-sparse solver).
//
// A pose-graph or general-graph solve needs ~70 small device arrays and ~30 small uploads; as ~70 hipMalloc + hipFree and
// ~30 pageable hipMemcpyAsync (each staged and waited for by the runtime) that set-up cost more host time than the kernels
// of a 400-keyframe loop closure took (GSLAM/core/Optimizer.h:127-148,162-167 is called once per loop closure: latency is
// what the caller sees).  Here the arrays are bump-allocated out of the context's grow-only graph arena (one hipMalloc
// per context and size class instead of one per array and solve), and the uploads are staged into the context's pinned
// block and sent with ONE DMA when they lie back to back in the arena (they do when they are allocated first).
//
// Two passes over the same allocation code: measure (sizes only, no memory is touched) -> reserve() -> the real pass.
// Arrays above kBigBytes (the dense keyframe system of a large graph) keep their own hipMalloc / hipFree: their
// allocation time is nothing beside their factorisation, and the arena would pin that much HBM for the life of the
// context.  GSLAM_HIP_PG_ARENA=0 gives every array its own hipMalloc and every upload its own copy (A/B measurements).
#pragma once
#include <stdlib.h>

#include <algorithm>

#include "common.h"

struct GraphArena {
  static constexpr size_t kBigBytes = (size_t)32 << 20;
  static constexpr size_t kStageCap = (size_t)8 << 20;  // most pinned host memory one flush() may ask the context for
  gh_ctx* ctx;
  bool measuring = true, enabled = true;
  size_t used = 0, want = 0;
  std::vector<void*> own;  // individually allocated (big, or the arena could not be grown)
  struct Piece {
    char* dst;
    const void* src;
    size_t bytes;
  };
  std::vector<Piece> pieces;
  struct Block {  // an array that lives in the arena, with the room it takes there
    const char* p;
    size_t room;
  };
  std::vector<Block> blocks;

  explicit GraphArena(gh_ctx* c) : ctx(c) {
    const char* e = getenv("GSLAM_HIP_PG_ARENA");
    enabled = !(e && e[0] == '0');
  }
  GraphArena(const GraphArena&) = delete;
  GraphArena& operator=(const GraphArena&) = delete;
  ~GraphArena() {
    if (own.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (void* p : own) (void)hipFree(p);
  }

  // 256-byte aligned, plus a 256-byte gap that belongs to nobody (an array of its own used to end in the slack of its
  // hipMalloc: keep that tolerance for a vector load that reaches a little past the last element)
  static size_t padded(size_t bytes) { return (((bytes ? bytes : 1) + 255) & ~(size_t)255) + 256; }

  template <typename T>
  bool alloc(T** out, size_t count) {
    const size_t bytes = padded(count * sizeof(T));
    const bool in_arena = enabled && bytes <= kBigBytes;
    if (measuring) {
      if (in_arena) want += bytes;
      *out = nullptr;
      return true;
    }
    if (in_arena && ctx->pg_arena && used + bytes <= ctx->pg_arena_bytes) {
      *out = reinterpret_cast<T*>(static_cast<char*>(ctx->pg_arena) + used);
      blocks.push_back(Block{static_cast<const char*>(ctx->pg_arena) + used, bytes});
      used += bytes;
      return true;
    }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    own.push_back(p);
    *out = static_cast<T*>(p);
    return true;
  }

  // After the measuring pass: make the arena hold `want` bytes (grow-only; nothing of an earlier solve lives in it -- every
  // solve ends with a stream synchronisation and holds the context's lock throughout), then switch to the real pass.
  gh_status reserve() {
    measuring = false;
    used = 0;
    if (!enabled || want <= ctx->pg_arena_bytes) return GH_OK;
    GH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->pg_arena) GH_HIP(ctx, hipFree(ctx->pg_arena));
    ctx->pg_arena = nullptr;
    ctx->pg_arena_bytes = 0;
    const size_t bytes = want + want / 4 + (1 << 20);  // graphs grow keyframe by keyframe: head-room for the next ones
    if (hipMalloc(&ctx->pg_arena, bytes) != hipSuccess) {
      (void)hipGetLastError();
      ctx->pg_arena = nullptr;  // every array takes the individual path
      return GH_OK;
    }
    ctx->pg_arena_bytes = bytes;
    return GH_OK;
  }

  // note a host -> device copy of `bytes` into `dst` (an array of this arena); sent by flush()
  void upload(void* dst, const void* src, size_t bytes) {
    if (bytes) pieces.push_back(Piece{static_cast<char*>(dst), src, bytes});
  }

  // Send the noted copies.  Pieces that follow each other in the arena (each starts where the padded previous one ends) form
  // a run: staged into the pinned block at the same relative offsets and sent as ONE copy (the padding between them
  // travels too: it belongs to nobody).  Whatever is left (own allocations, or no pinned memory) goes piece by piece.
  // The staging block is the context's (gh_pinned): it must not be asked for again before the stream has been
  // synchronised -- flush() does not wait.
  gh_status flush() {
    size_t i = 0;
    char* stage = nullptr;
    size_t stage_used = 0, stage_bytes = 0;
    // room of the arena array that starts at p (0: not an arena array -- two hipMalloc'ed arrays may happen to be
    // neighbours, but one copy must not span both)
    auto room = [&](const char* p) -> size_t {
      for (const Block& b : blocks)
        if (b.p == p) return b.room;
      return 0;
    };
    if (enabled) {
      for (const Piece& p : pieces) stage_bytes += std::max(room(p.dst), padded(p.bytes));
      void* hp = nullptr;
      // (a graph with millions of observations uploads hundreds of MB: not worth pinning that much host memory for,
      // and its set-up time is not what its caller waits for)
      // (and the pinned block is grow-only: more than kStageCap is sent piece by piece straight from the caller's memory)
      if (stage_bytes && stage_bytes <= kStageCap && gh_pinned(ctx, stage_bytes, &hp) == GH_OK) stage = static_cast<char*>(hp);
    }
    while (i < pieces.size()) {
      size_t j = i + 1;
      if (stage) {
        for (; j < pieces.size(); ++j) {
          const size_t r = room(pieces[j - 1].dst);
          if (r == 0 || room(pieces[j].dst) == 0 || pieces[j].dst != pieces[j - 1].dst + r) break;
        }
        char* s0 = stage + stage_used;
        for (size_t k = i; k < j; ++k) memcpy(s0 + (pieces[k].dst - pieces[i].dst), pieces[k].src, pieces[k].bytes);
        const size_t span = (size_t)(pieces[j - 1].dst - pieces[i].dst) + pieces[j - 1].bytes;
        GH_HIP(ctx, hipMemcpyAsync(pieces[i].dst, s0, span, hipMemcpyHostToDevice, ctx->stream));
        stage_used += padded(span);
      } else {
        GH_HIP(ctx, hipMemcpyAsync(pieces[i].dst, pieces[i].src, pieces[i].bytes, hipMemcpyHostToDevice, ctx->stream));
      }
      i = j;
    }
    pieces.clear();
    return GH_OK;
  }
};
