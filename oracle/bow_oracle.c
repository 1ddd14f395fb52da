/*
 * bow_oracle.c — CPU restatement of GSLAM::Vocabulary's BoW transform.  TEST INFRASTRUCTURE ONLY.
 * PINNED: tests/test_bow_oracle.py compares this file, output for output, with the reference's own
 * Vocabulary::load + transform compiled from /root/reference into oracle/_ref (ref_vocab_*), and with the committed
 * golden vectors tests/golden/bow_reference.npz generated from it.
 *
 * Follows:
 *   GSLAM/core/Vocabulary.h:1695-1736  transform(feature, word_id, weight, nid, levelsup): greedy descent from
 *                                      the root; children of node p are p*k+1 .. p*k+childNum (:1716); strict '<'
 *                                      keeps the FIRST minimum (:1719); stops at a node with childNum == 0;
 *                                      nid = node reached at level L - levelsup (root if that level <= 0)
 *   GSLAM/core/Vocabulary.h:485-491    distance = hamming32
 *   GSLAM/core/Vocabulary.h:1558-1621  image transform: TF / TF_IDF accumulate v[id] += w in feature order
 *                                      (float), IDF / BINARY keep the first w; words with w <= 0 are skipped;
 *                                      without normalisation TF-type values are divided by the vector size
 *   GSLAM/core/Vocabulary.h:386-408    normalize: L1 = sum |v| (double, ascending word id), L2 = sqrt(sum v^2);
 *                                      v /= norm (float /= double)
 *   GSLAM/core/Vocabulary.h:667-683    which scoring types normalise (all L1 except L2_NORM = L2, DOT_PRODUCT = none)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int oracle_hamming32(const uint8_t* a, const uint8_t* b);

typedef struct {
  uint32_t childNum;
  float weight;
} bow_node;

/* Vocabulary.h:493-513: hamming64 / hamming8x = popcount of the xor over the descriptor, 64 bits at a time */
static int hamming_bytes(const uint8_t* a, const uint8_t* b, int bytes) {
  int d = 0;
  for (int w = 0; w < bytes / 8; ++w) {
    uint64_t x, y;
    memcpy(&x, a + 8 * w, 8);
    memcpy(&y, b + 8 * w, 8);
    d += __builtin_popcountll(x ^ y);
  }
  return d;
}

void oracle_bow_word_bytes(const bow_node* nodes, const uint8_t* ndesc, int k, int L, const uint8_t* f, int levelsup,
                           uint32_t* word, float* weight, uint32_t* node, int desc_bytes);

void oracle_bow_word(const bow_node* nodes, const uint8_t* ndesc, int k, int L, const uint8_t* f, int levelsup,
                     uint32_t* word, float* weight, uint32_t* node) {
  oracle_bow_word_bytes(nodes, ndesc, k, L, f, levelsup, word, weight, node, 32);
}

/* Vocabulary.h:550-560 l2generic: squared L2, float accumulation in index order (no FMA: built with -ffp-contract=off) */
static float l2_f32(const float* a, const float* b, int dims) {
  float sqd = 0.f;
  for (int i = 0; i < dims; ++i) {
    const float tmp = a[i] - b[i];
    sqd += tmp * tmp;
  }
  return sqd;
}

/* float vocabulary: same descent with l2generic, FLT_MAX start, first strict minimum (Vocabulary.h:1712-1725) */
void oracle_bow_word_f32(const bow_node* nodes, const float* ndesc, int k, int L, const float* f, int levelsup, uint32_t* word,
                         float* weight, uint32_t* node, int dims) {
  const int nid_level = L - levelsup;
  uint32_t final_id = 0, nid = 0;
  int level = 0;
  do {
    ++level;
    float best_d = 3.402823466e+38f;
    uint32_t best = final_id;
    uint32_t id = final_id * (uint32_t)k + 1;
    for (uint32_t end = id + nodes[final_id].childNum; id < end; ++id) {
      const float d = l2_f32(f, ndesc + (size_t)id * dims, dims);
      if (d < best_d) {
        best_d = d;
        best = id;
      }
    }
    if (best == final_id) break;
    final_id = best;
    if (level == nid_level) nid = final_id;
  } while (nodes[final_id].childNum != 0);
  *word = final_id;
  *weight = nodes[final_id].weight;
  *node = nid_level <= 0 ? 0 : nid;
}

/* desc_bytes: any multiple of 8 (DistanceFactory::create, Vocabulary.h:560-568) */
void oracle_bow_word_bytes(const bow_node* nodes, const uint8_t* ndesc, int k, int L, const uint8_t* f, int levelsup,
                           uint32_t* word, float* weight, uint32_t* node, int desc_bytes) {
  const int nid_level = L - levelsup;
  uint32_t final_id = 0, nid = 0;
  int level = 0;
  do {
    ++level;
    int best_d = 1 << 30;
    uint32_t best = final_id;
    uint32_t id = final_id * (uint32_t)k + 1;
    for (uint32_t end = id + nodes[final_id].childNum; id < end; ++id) {
      int d = desc_bytes == 32 ? oracle_hamming32(f, ndesc + (size_t)id * 32) : hamming_bytes(f, ndesc + (size_t)id * desc_bytes, desc_bytes);
      if (d < best_d) {
        best_d = d;
        best = id;
      }
    }
    final_id = best;
    if (level == nid_level) nid = final_id;
  } while (nodes[final_id].childNum != 0);
  *word = final_id;
  *weight = nodes[final_id].weight;
  *node = nid_level <= 0 ? 0 : nid;
}

static int cmp_u32(const void* a, const void* b) {
  uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; scoring: 0 L1, 1 L2, 2 CHI2, 3 KL, 4 BHATT, 5 DOT.
 * Outputs per feature (word, weight, node) and the BoW vector (ascending word id).  Returns the BoW length. */
int oracle_bow_transform_bytes(const bow_node* nodes, const uint8_t* ndesc, int k, int L, int weighting, int scoring,
                               const uint8_t* desc, int n, int levelsup, uint32_t* word, float* weight, uint32_t* node,
                               uint32_t* bow_word, float* bow_val, int desc_bytes);

int oracle_bow_transform(const bow_node* nodes, const uint8_t* ndesc, int k, int L, int weighting, int scoring,
                         const uint8_t* desc, int n, int levelsup, uint32_t* word, float* weight, uint32_t* node,
                         uint32_t* bow_word, float* bow_val) {
  return oracle_bow_transform_bytes(nodes, ndesc, k, L, weighting, scoring, desc, n, levelsup, word, weight, node, bow_word,
                                    bow_val, 32);
}

int oracle_bow_transform_bytes(const bow_node* nodes, const uint8_t* ndesc, int k, int L, int weighting, int scoring,
                               const uint8_t* desc, int n, int levelsup, uint32_t* word, float* weight, uint32_t* node,
                               uint32_t* bow_word, float* bow_val, int desc_bytes) {
  /* desc_bytes < 0: float descriptors of -desc_bytes / 4 dimensions */
  for (int i = 0; i < n; ++i) {
    if (desc_bytes < 0)
      oracle_bow_word_f32(nodes, (const float*)ndesc, k, L, (const float*)(desc + (size_t)i * (size_t)(-desc_bytes)), levelsup,
                          &word[i], &weight[i], &node[i], -desc_bytes / 4);
    else
      oracle_bow_word_bytes(nodes, ndesc, k, L, desc + (size_t)i * desc_bytes, levelsup, &word[i], &weight[i], &node[i], desc_bytes);
  }
  /* stable order by word id: features of one word keep their order (only counts matter: same weight) */
  uint32_t* ids = (uint32_t*)malloc(sizeof(uint32_t) * (n > 0 ? n : 1));
  int m = 0;
  for (int i = 0; i < n; ++i)
    if (weight[i] > 0) ids[m++] = word[i];
  qsort(ids, m, sizeof(uint32_t), cmp_u32);
  int nb = 0;
  for (int i = 0; i < m;) {
    int j = i;
    while (j < m && ids[j] == ids[i]) ++j;
    float w = nodes[ids[i]].weight, v = w;
    if (weighting == 0 || weighting == 1)
      for (int c = 1; c < j - i; ++c) v += w; /* addWeight: repeated float += in feature order */
    bow_word[nb] = ids[i];
    bow_val[nb] = v;
    ++nb;
    i = j;
  }
  free(ids);
  const int must = scoring != 5;
  if ((weighting == 0 || weighting == 1) && nb > 0 && !must) {
    const double nd = (double)nb;
    for (int i = 0; i < nb; ++i) bow_val[i] = (float)(bow_val[i] / nd);
  }
  if (must) {
    double norm = 0.0;
    if (scoring == 1) {
      for (int i = 0; i < nb; ++i) norm += bow_val[i] * bow_val[i]; /* float * float, accumulated in double */
      norm = sqrt(norm);
    } else {
      for (int i = 0; i < nb; ++i) norm += fabs(bow_val[i]);
    }
    if (norm > 0.0)
      for (int i = 0; i < nb; ++i) bow_val[i] = (float)(bow_val[i] / norm);
  }
  return nb;
}

/* L1 score of two BoW vectors (Vocabulary.h:691-736): s = sum over common words of |vi - wi| - |vi| - |wi|; -s/2 */
double oracle_bow_score_l1(const uint32_t* a_id, const float* a_v, int na, const uint32_t* b_id, const float* b_v,
                           int nb) {
  double score = 0;
  int i = 0, j = 0;
  while (i < na && j < nb) {
    if (a_id[i] == b_id[j]) {
      const float vi = a_v[i], wi = b_v[j];
      /* the reference's unqualified fabs() on floats resolves to the float overload (C++ <cmath>): the three
       * terms are combined in single precision, then accumulated in double (checked against oracle/_ref) */
      const float term = fabsf(vi - wi) - fabsf(vi) - fabsf(wi);
      score += term;
      ++i;
      ++j;
    } else if (a_id[i] < b_id[j]) {
      ++i;
    } else {
      ++j;
    }
  }
  return -score / 2.0;
}

/* All six scoring classes of the reference (GSLAM/core/Vocabulary.h:691-979), selected by the ScoringType enum order
 * (L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT = 0..5); a = v1, b = v2, both ascending by word id.
 * WordValue is float and the reference calls fabs / sqrt / log unqualified on floats, which C++ <cmath> resolves to the
 * FLOAT overloads: every per-word term is formed in single precision and accumulated into a double, in ascending id
 * order.  Checked bit for bit against oracle/_ref (tests/test_bow_oracle.py). */
static const double BOW_LOG_EPS = -36.043653389117154; /* log(DBL_EPSILON), GeneralScoring::LOG_EPS (:631-634) */

double oracle_bow_score(int scoring, const uint32_t* a_id, const float* a_v, int na, const uint32_t* b_id,
                        const float* b_v, int nb) {
  double score = 0;
  int i = 0, j = 0;
  if (scoring == 3) { /* KLScoring (:843-891): every word of v1 contributes */
    while (i < na && j < nb) {
      const float vi = a_v[i], wi = b_v[j];
      if (a_id[i] == b_id[j]) {
        if (vi != 0 && wi != 0) score += vi * logf(vi / wi);
        ++i;
        ++j;
      } else if (a_id[i] < b_id[j]) {
        score += vi * (logf(vi) - BOW_LOG_EPS);
        ++i;
      } else {
        ++j; /* lower_bound(v1 id): no contribution */
      }
    }
    for (; i < na; ++i)
      if (a_v[i] != 0) score += a_v[i] * (logf(a_v[i]) - BOW_LOG_EPS);
    return score;
  }
  while (i < na && j < nb) {
    if (a_id[i] == b_id[j]) {
      const float vi = a_v[i], wi = b_v[j];
      switch (scoring) {
        case 0: score += fabsf(vi - wi) - fabsf(vi) - fabsf(wi); break;         /* L1 (:691-736) */
        case 1: case 5: score += vi * wi; break;                                /* L2 (:741-790), dot (:939-979) */
        case 2: if (vi + wi != 0.0) score += vi * wi / (vi + wi); break;        /* chi square (:795-838) */
        case 4: score += sqrtf(vi * wi); break;                                 /* Bhattacharyya (:896-934) */
      }
      ++i;
      ++j;
    } else if (a_id[i] < b_id[j]) {
      ++i;
    } else {
      ++j;
    }
  }
  switch (scoring) {
    case 0: return -score / 2.0;
    case 1: return score >= 1 ? 1.0 : 1.0 - sqrt(1.0 - score);
    case 2: return 2. * score;
    default: return score;
  }
}
