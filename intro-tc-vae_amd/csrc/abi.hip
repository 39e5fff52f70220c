// Library-level entry points of libitcv_hip.so (error reporting, ABI version).
#include "common.h"

namespace itcv {
thread_local char g_err[512] = "";
}

extern "C" {
int itcv_abi_version(void) { return ITCV_ABI_VERSION; }
const char* itcv_last_error(void) { return itcv::g_err; }
}
