/* ORACLE (test infrastructure, not product code) -- plain-C restatement.
 *
 * A second, independent CPU restatement of the reference path, used (a) to
 * cross-check oracle/attention.py and oracle/side_inputs.py and (b) as the
 * scalar "port" the bench can time.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.
 *
 * PARITY: the integer generators are pinned by the golden matrices of
 * src/feature_utils_test.py:64-72,95-108 (tests/golden/).  The float operator is
 * PARITY UNPINNED: etcmodel (un-vendored, version-less, src/README.md:10-11) is
 * absent and the reference has no attention fixtures; see oracle/attention.py.
 *
 * Follows: SURVEY.md App. A.2 / A.3;  src/feature_utils.py:89-184;
 *          src/data/data_utils.py:350-368; call site mmt_encoder.py:220-224.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* 1-D clipped relative id (etcmodel RelativePositionGenerator; App. A.2). */
static inline int32_t id_1d(int i, int j, int m) {
  int d = j - i;
  if (d >= 0) return d < m ? d : m;
  d = -d;
  return m + (d < m ? d : m);
}

/* 2-D id of key patch (xk,yk) seen from query patch (xq,yq): the value of the
 * reference's base tensor at [P - xq + xk, P - yq + yk] (feature_utils.py:164-170),
 * evaluated in closed form from the region layout of :89-112 / :221-254. */
static inline int32_t id_2d(int dx, int dy, int r) {
  int d = 2 * r + 1;
  int vert = dx < -r ? -1 : (dx > r ? 1 : 0);   /* -1 = above the core rows */
  int horz = dy < -r ? -1 : (dy > r ? 1 : 0);
  if (vert == 0 && horz == 0) {
    int c = dx * d + dy;                          /* roll(arange(d*d), d*r + r) */
    return c < 0 ? c + d * d : c;
  }
  /* ids d*d + {top, top_right, right, right_bottom, bottom, bottom_left, left, top_left} */
  static const int8_t dir[3][3] = { {7, 0, 1}, {6, -1, 2}, {5, 4, 3} };
  return d * d + dir[vert + 1][horz + 1];
}

void oracle_relative_ids(int32_t* out, int S, int id_mode, int max_dist,
                         int P, int r) {
  int I = (id_mode == 2) ? P * P : 0;
  int image_part = P * P + 8 + (2 * max_dist + 1);   /* feature_utils.py:78-79 */
  int text_part = image_part + 1;                     /* :82 */
  for (int i = 0; i < S; ++i)
    for (int j = 0; j < S; ++j) {
      int32_t v;
      if (id_mode == 1) v = id_1d(i, j, max_dist);
      else if (i < I && j < I) v = id_2d(j / P - i / P, j % P - i % P, r);
      else if (i < I) v = text_part;
      else if (j < I) v = image_part;
      else v = id_1d(i, j, max_dist);
      out[(size_t)i * S + j] = v;
    }
}

/* att_mask: segmented (valid x valid, pad x pad) optionally intersected with the
 * build-defined band + global pattern (App. A.5); radius < 0 means "no band". */
void oracle_att_mask(int32_t* out, int S, int valid_len, int radius,
                     int global_start, int n_global) {
  for (int i = 0; i < S; ++i)
    for (int j = 0; j < S; ++j) {
      int seg = (i < valid_len) == (j < valid_len);
      int ok = seg;
      if (radius >= 0) {
        int d = i > j ? i - j : j - i;
        int gi = i >= global_start && i < global_start + n_global;
        int gj = j >= global_start && j < global_start + n_global;
        ok = seg && (d <= radius || gi || gj);
      }
      out[(size_t)i * S + j] = ok;
    }
}

void oracle_segment_ids(int32_t* out, int S, int img_wp, int txt_wp) {
  for (int p = 0; p < S; ++p)
    out[p] = (p < img_wp ? 1 : 0) + ((p > img_wp && p < img_wp + txt_wp) ? 2 : 0);
}

/* Dense QkvRelativeAttention forward, one thread, fp32 storage with fp32
 * accumulation in the reference's op order (acc64 != 0: double accumulators).
 * q,k,v,out: [B,S,N,D]; rel_emb [R,N,D]; rel_bias [R,N] or NULL; att_mask and
 * rel_ids [B,S,S] (batch stride 0 allowed via mask_bstride/ids_bstride). */
int oracle_rel_attention_fwd(const float* q, const float* k, const float* v,
                             const float* rel_emb, const float* rel_bias,
                             const int32_t* att_mask, const int32_t* rel_ids,
                             long mask_bstride, long ids_bstride,
                             float* out, float* lse,
                             int B, int S, int N, int D, int R,
                             float scale, float mask_value, int scale_after_add,
                             int acc64) {
  float* s = (float*)malloc(sizeof(float) * (size_t)S);
  float* relall = (float*)malloc(sizeof(float) * (size_t)(R > 0 ? R : 1));
  if (!s || !relall) { free(s); free(relall); return -1; }
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n)
      for (int i = 0; i < S; ++i) {
        const float* qi = q + (((size_t)b * S + i) * N + n) * D;
        for (int r = 0; r < R; ++r) {
          const float* e = rel_emb + ((size_t)r * N + n) * D;
          if (acc64) { double a = 0; for (int d = 0; d < D; ++d) a += (double)qi[d] * e[d];
                       relall[r] = (float)a; }
          else { float a = 0; for (int d = 0; d < D; ++d) a += qi[d] * e[d]; relall[r] = a; }
          if (rel_bias) relall[r] += rel_bias[(size_t)r * N + n];
        }
        float mx = -INFINITY;
        for (int j = 0; j < S; ++j) {
          const float* kj = k + (((size_t)b * S + j) * N + n) * D;
          float c;
          if (acc64) { double a = 0; for (int d = 0; d < D; ++d) a += (double)qi[d] * kj[d];
                       c = (float)a; }
          else { float a = 0; for (int d = 0; d < D; ++d) a += qi[d] * kj[d]; c = a; }
          if (!scale_after_add) c *= scale;
          if (rel_ids && R > 0) {
            int32_t id = rel_ids[(size_t)b * ids_bstride + (size_t)i * S + j];
            if (id >= 0 && id < R) c += relall[id];
          }
          if (scale_after_add) c *= scale;
          if (att_mask)
            c += (float)(1 - att_mask[(size_t)b * mask_bstride + (size_t)i * S + j]) * mask_value;
          s[j] = c;
          if (c > mx) mx = c;
        }
        double den = 0;
        for (int j = 0; j < S; ++j) { s[j] = expf(s[j] - mx); den += s[j]; }
        float inv = (float)(1.0 / den);
        float* oi = out + (((size_t)b * S + i) * N + n) * D;
        for (int d = 0; d < D; ++d) {
          double a = 0;
          for (int j = 0; j < S; ++j)
            a += (double)(s[j] * inv) * v[(((size_t)b * S + j) * N + n) * D + d];
          oi[d] = (float)a;
        }
        if (lse) lse[((size_t)b * N + n) * S + i] = mx + (float)log(den);
      }
  free(s); free(relall);
  return 0;
}
