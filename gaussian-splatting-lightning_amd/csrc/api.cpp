// api.cpp — ABI version and thread-local error text of libgspl_hip.so (see include/gspl_hip.h).
#include "gspl_host.h"
#include <cstdio>

namespace gspl {
static thread_local char g_err[512] = "no error";
void set_error(const char* where, const char* what) {
    std::snprintf(g_err, sizeof(g_err), "%s: %s", where ? where : "?", what ? what : "?");
}
}  // namespace gspl

extern "C" int gspl_abi_version(void) { return 35; }
extern "C" const char* gspl_last_error(void) { return gspl::g_err; }
extern "C" size_t gspl_inria_state_bytes(void) { return sizeof(gspl_inria_state); }
