// Thread-local error channel shared by the C-ABI translation units (mmt_last_error()).
#pragma once
namespace mmt {
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
}
