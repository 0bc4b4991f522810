// Thread-local error channel shared by the C-ABI translation units (mmt_last_error()).
#pragma once
namespace mmt {
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// mmt_set_step_scalars (mmt_attn.h): device-resident per-step scalars, copied into every launch's parameters
extern const unsigned long long* g_dropout_epoch;
extern const float* g_adamw_hyper;
}
