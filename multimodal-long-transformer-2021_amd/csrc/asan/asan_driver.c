/* C driver of the sanitizer build (`make asan`, CPU only): descriptor validation, error channel and workspace
 * sizing of the C-ABI shim (mmt_api.hip, host side) under -fsanitize=address,undefined.  The kernel launchers are
 * the stand-ins of asan_stubs.cpp: nothing touches a GPU. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../../include/mmt_attn.h"

extern const unsigned char* g_ws_lo;
extern const unsigned char* g_ws_hi;
extern int g_launches, g_last_kind, g_last_handover;
extern const void* g_last_epoch;

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "asan driver: %s:%d: %s (last error: %s)\n", __FILE__, __LINE__, #cond, mmt_last_error()); return 1; } } while (0)

static mmt_attn_desc base_desc(int B, int S, int N, int R, int dtype) {
  mmt_attn_desc d;
  memset(&d, 0, sizeof(d));
  d.B = B; d.S = S; d.N = N; d.D = 64; d.R = R; d.dtype = dtype;
  int64_t st[3] = {(int64_t)S * N * 64, (int64_t)N * 64, 64};
  for (int i = 0; i < 3; ++i) d.q_stride[i] = d.k_stride[i] = d.v_stride[i] = d.o_stride[i] = st[i];
  d.scale = 0.125f; d.mask_value = -10000.f;
  d.mask.local_radius = 64; d.mask.id_mode = R ? MMT_IDS_1D : MMT_IDS_NONE; d.mask.max_dist = 12;
  return d;
}

int main(void) {
  char dummy[64];
  CHECK(mmt_abi_version() == MMT_ABI_VERSION);
  /* ---- argument errors: codes + message, no launch ---- */
  CHECK(mmt_attn_fwd(NULL, dummy, dummy, dummy, NULL, NULL, NULL, NULL, dummy, NULL, NULL, 0, NULL) == MMT_E_INVALID);
  CHECK(strstr(mmt_last_error(), "desc is NULL") != NULL);
  mmt_attn_desc d = base_desc(2, 300, 3, 32, MMT_BF16);
  d.D = 32;
  CHECK(mmt_attn_fwd(&d, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, NULL, 0, NULL) == MMT_E_UNSUPPORTED);
  CHECK(mmt_workspace_bytes(&d) == 0);
  d = base_desc(2, 300, 3, 129, MMT_BF16);      /* tables are built up to 128 ids wide */
  CHECK(mmt_attn_fwd(&d, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, NULL, 0, NULL) == MMT_E_UNSUPPORTED);
  d = base_desc(2, 300, 3, 32, MMT_BF16);
  d.q_stride[1] = 7;                         /* not a multiple of 8 elements */
  CHECK(mmt_attn_fwd(&d, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, NULL, 0, NULL) == MMT_E_INVALID);
  d = base_desc(2, 300, 3, 32, MMT_BF16);
  d.mask.global_start = 290; d.mask.n_global = 20;   /* range past the sequence */
  CHECK(mmt_attn_fwd(&d, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, NULL, 0, NULL) == MMT_E_INVALID);
  d = base_desc(2, 300, 3, 32, MMT_BF16);
  d.dropout_p = 1.0f;
  CHECK(mmt_attn_fwd(&d, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, NULL, 0, NULL) == MMT_E_INVALID);
  d = base_desc(2, 300, 3, 32, MMT_BF16);
  CHECK(mmt_attn_fwd(&d, NULL, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, NULL, 0, NULL) == MMT_E_INVALID);
  CHECK(mmt_attn_fwd(&d, dummy, dummy, dummy, NULL, NULL, NULL, NULL, dummy, NULL, NULL, 0, NULL) == MMT_E_INVALID);   /* R > 0, no table */
  int32_t gidx[3] = {3, 9, 200};
  d.mask.global_index = gidx; d.mask.n_global = 3;
  CHECK(mmt_attn_fwd(&d, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, NULL, 0, NULL) == MMT_E_UNSUPPORTED);
  CHECK(strstr(mmt_last_error(), "listed global-token set") != NULL);
  CHECK(g_launches == 0);

  /* ---- workspace sizing: every derived pointer stays inside what mmt_workspace_bytes asked for ---- */
  static const int shapes[][6] = {   /* B, S, N, R, n_global, dtype */
      {1, 20, 1, 9, 1, MMT_F32},  {2, 300, 3, 32, 8, MMT_BF16}, {1, 1024, 2, 32, 8, MMT_F32},
      {4, 4096, 12, 32, 8, MMT_BF16}, {2, 8192, 12, 32, 128, MMT_BF16}, {1, 513, 5, 49, 40, MMT_BF16},
      {1, 96, 1, 0, 0, MMT_BF16}};
  for (unsigned i = 0; i < sizeof(shapes) / sizeof(shapes[0]); ++i) {
    const int* s = shapes[i];
    d = base_desc(s[0], s[1], s[2], s[3], s[5]);
    d.mask.n_global = s[4]; d.mask.global_start = s[4] ? s[1] / 2 - s[4] / 2 : 0;
    const size_t need = mmt_workspace_bytes(&d);
    CHECK(need > 0 || s[4] == 0);
    unsigned char* ws = (unsigned char*)malloc(need ? need : 1);    /* exact size: an overrun is an ASan report */
    CHECK(ws != NULL);
    g_ws_lo = ws; g_ws_hi = ws + need;
    const int before = g_launches;
    if (need > 16) {   /* one byte short is refused, with the requirement in the message */
      CHECK(mmt_attn_bwd(&d, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, dummy, (const float*)dummy, dummy, dummy, dummy,
                         (float*)dummy, NULL, ws, need - 1, NULL) == MMT_E_WORKSPACE);
      CHECK(g_launches == before);
    }
    CHECK(mmt_attn_fwd(&d, dummy, dummy, dummy, s[3] ? dummy : NULL, NULL, NULL, NULL, dummy, (float*)dummy, ws, need, NULL) == MMT_OK);
    CHECK(mmt_attn_bwd(&d, dummy, dummy, dummy, s[3] ? dummy : NULL, NULL, NULL, NULL, dummy, dummy, (const float*)dummy, dummy, dummy, dummy,
                       s[3] ? (float*)dummy : NULL, NULL, ws, need, NULL) == MMT_OK);
    CHECK(g_launches > before);
    free(ws);
  }

  /* ---- side inputs: the reference generator's argument errors (feature_utils.py:60-65) ---- */
  mmt_mask_desc m;
  memset(&m, 0, sizeof(m));
  m.id_mode = MMT_IDS_2D; m.patches_per_row = 0; m.core_layers = 1; m.max_dist = 3;
  CHECK(mmt_side_inputs(&m, 1, 64, NULL, NULL, 0, NULL, (int32_t*)dummy, NULL, NULL) == MMT_E_INVALID);
  CHECK(strstr(mmt_last_error(), "num_patch_per_row") != NULL);
  m.patches_per_row = 4; m.core_layers = 0;
  CHECK(mmt_side_inputs(&m, 1, 64, NULL, NULL, 0, NULL, (int32_t*)dummy, NULL, NULL) == MMT_E_INVALID);
  m.core_layers = 1; m.max_dist = -1;
  CHECK(mmt_side_inputs(&m, 1, 64, NULL, NULL, 0, NULL, (int32_t*)dummy, NULL, NULL) == MMT_E_INVALID);
  m.max_dist = 3;
  CHECK(mmt_side_inputs(&m, 1, 8, NULL, NULL, 0, NULL, (int32_t*)dummy, NULL, NULL) == MMT_E_INVALID);   /* image longer than S */
  CHECK(mmt_side_inputs(&m, 1, 64, NULL, NULL, 0, NULL, (int32_t*)dummy, NULL, NULL) == MMT_OK && g_last_kind == 6);
  CHECK(mmt_side_inputs(NULL, 1, 64, NULL, NULL, 0, NULL, NULL, NULL, NULL) == MMT_E_INVALID);
  /* ---- per-step scalars in device memory (ABI 4: named by the descriptors, no registration in the library) ---- */
  CHECK(mmt_write_step_scalars(NULL, NULL, 1, 1e-4f, 0.1f, 0.001f, NULL) == MMT_E_INVALID);
  CHECK(mmt_write_step_scalars((uint64_t*)dummy, (float*)dummy, 1, 1e-4f, 0.1f, 0.001f, NULL) == MMT_OK && g_last_kind == 7);
  {   /* two descriptors with different epoch words in one process: each launch gets its own */
    uint64_t epoch_a = 1, epoch_b = 2;
    mmt_attn_desc da = base_desc(1, 256, 2, 32, MMT_BF16), db = base_desc(1, 256, 2, 32, MMT_BF16);
    da.dropout_p = db.dropout_p = 0.1f;
    da.dropout_epoch = &epoch_a; db.dropout_epoch = &epoch_b;
    unsigned char* ws = (unsigned char*)malloc(mmt_workspace_bytes(&da) + 1);
    g_ws_lo = ws; g_ws_hi = ws + mmt_workspace_bytes(&da) + 1;
    CHECK(mmt_attn_fwd(&da, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, ws, mmt_workspace_bytes(&da), NULL) == MMT_OK);
    CHECK(g_last_epoch == (const void*)&epoch_a);
    CHECK(mmt_attn_fwd(&db, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, ws, mmt_workspace_bytes(&db), NULL) == MMT_OK);
    CHECK(g_last_epoch == (const void*)&epoch_b);
    CHECK(mmt_attn_bwd(&da, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, dummy, (const float*)dummy, dummy, dummy, dummy,
                       (float*)dummy, NULL, ws, mmt_workspace_bytes(&da), NULL) == MMT_OK);
    CHECK(g_last_epoch == (const void*)&epoch_a);
    db.dropout_p = 0.f;                       /* no dropout: the word is not even passed on */
    CHECK(mmt_attn_fwd(&db, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, ws, mmt_workspace_bytes(&db), NULL) == MMT_OK);
    CHECK(g_last_epoch == NULL);
    /* kernel-selection switches travel in the descriptor too (they were environment variables until ABI 3) */
    da.tuning = MMT_TUNE_FWD_NO_WIN;
    CHECK(mmt_attn_fwd(&da, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, ws, mmt_workspace_bytes(&da), NULL) == MMT_OK && g_last_kind == 2);
    da.tuning = MMT_TUNE_FWD_FORCE_WIN;
    CHECK(mmt_attn_fwd(&da, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, ws, mmt_workspace_bytes(&da), NULL) == MMT_OK && g_last_kind == 3);
    {   /* the plane-walk forward needs the caller's arrival counters when there are global tokens (ABI 4: desc.sync) */
      mmt_attn_desc dw = base_desc(4, 4096, 12, 32, MMT_BF16);
      dw.mask.n_global = 8; dw.mask.global_start = 3971;
      const size_t need = mmt_workspace_bytes(&dw);
      unsigned char* w2 = (unsigned char*)malloc(need);
      g_ws_lo = w2; g_ws_hi = w2 + need;
      CHECK(mmt_attn_fwd(&dw, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, w2, need, NULL) == MMT_OK && g_last_kind == 3);   /* default: window kernel */
      dw.tuning = MMT_TUNE_FWD_WALK;                             /* asked for, but no counters: still the window kernel */
      CHECK(mmt_attn_fwd(&dw, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, w2, need, NULL) == MMT_OK && g_last_kind == 3);
      dw.sync = (uint32_t*)dummy; dw.sync_words = 47;            /* too few words for 48 planes */
      CHECK(mmt_attn_fwd(&dw, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, w2, need, NULL) == MMT_OK && g_last_kind == 3);
      dw.sync_words = 48;
      CHECK(mmt_attn_fwd(&dw, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, w2, need, NULL) == MMT_OK && g_last_kind == 8);
      dw.tuning = MMT_TUNE_FWD_PWIN;                             /* likewise opt-in: the sliding-window kernel */
      CHECK(mmt_attn_fwd(&dw, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, w2, need, NULL) == MMT_OK && g_last_kind == 9);
      dw.tuning = 0;                                             /* the default stays the window kernel, counters or not */
      CHECK(mmt_attn_fwd(&dw, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, NULL, w2, need, NULL) == MMT_OK && g_last_kind == 3);
      free(w2);
      g_ws_lo = ws; g_ws_hi = ws + mmt_workspace_bytes(&da) + 1;
    }
    da.tuning = MMT_TUNE_BWD_NO_HANDOVER;
    CHECK(mmt_attn_bwd(&da, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, dummy, (const float*)dummy, dummy, dummy, dummy,
                       (float*)dummy, NULL, ws, mmt_workspace_bytes(&da), NULL) == MMT_OK && g_last_handover == 0);
    da.tuning = 0;
    CHECK(mmt_attn_bwd(&da, dummy, dummy, dummy, dummy, NULL, NULL, NULL, dummy, dummy, (const float*)dummy, dummy, dummy, dummy,
                       (float*)dummy, NULL, ws, mmt_workspace_bytes(&da), NULL) == MMT_OK && g_last_handover == 1);
    free(ws);
  }
  printf("asan driver ok: %d stand-in launches, no sanitizer report\n", g_launches);
  return 0;
}
