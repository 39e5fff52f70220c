// Library-level entry points of libitcv_hip.so (error reporting, ABI version).
#include "common.h"

#include <signal.h>
#include <unistd.h>

#include <vector>

namespace itcv {
thread_local char g_err[512] = "";

struct ProfRec {
  int code;
  double flop;
  hipEvent_t start, stop;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;

ProfScope::ProfScope(hipStream_t stream, int kind, int ks, int bm, int up2, int ns, double flop) : st(stream), slot(-1) {
  if (!g_prof_on) return;
  ProfRec r;
  r.code = kind | (ks << 4) | (bm << 8) | (up2 << 16) | (ns << 20);
  r.flop = flop;
  if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
  g_prof.push_back(r);
  slot = (int)g_prof.size() - 1;
  g_prof_start = r.start, g_prof_stop = r.stop;   // consumed by the scope's launch_timed()
}
ProfScope::~ProfScope() {
  if (slot >= 0 && g_prof_start) {   // nothing was launched through launch_timed(): drop the record
    (void)hipEventDestroy(g_prof[slot].start);
    (void)hipEventDestroy(g_prof[slot].stop);
    g_prof.pop_back();
  }
  g_prof_start = g_prof_stop = nullptr;
}
thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;
}

namespace itcv {
// Last words of a measurement process (bench.py, N>1 captured leg): a failure inside RCCL's watchdog thread ends in
// abort(), which no Python-level handler survives.  Armed, SIGABRT writes the prepared line (the finished eager
// measurement, marked "graph_leg": abandoned) to stdout with write(2) and leaves with _exit(0).
static char g_abort_line[65536];
static size_t g_abort_len = 0;
static void abort_line_handler(int) {
  size_t off = 0;
  while (off < g_abort_len) {
    const ssize_t r = write(1, g_abort_line + off, g_abort_len - off);
    if (r <= 0) break;
    off += (size_t)r;
  }
  _exit(0);
}
}  // namespace itcv

extern "C" {
int itcv_on_abort_print(const char* line) {
  using namespace itcv;
  if (!line) {                               // disarm
    g_abort_len = 0;
    signal(SIGABRT, SIG_DFL);
    return 0;
  }
  const size_t n = strlen(line);
  if (n + 2 > sizeof(g_abort_line)) return fail("%s: line too long (%lld bytes)", "itcv_on_abort_print", (long long)n);
  memcpy(g_abort_line, line, n);
  g_abort_len = n;
  if (n && g_abort_line[n - 1] != '\n') g_abort_line[g_abort_len++] = '\n';
  signal(SIGABRT, abort_line_handler);
  return 0;
}

int itcv_abi_version(void) { return ITCV_ABI_VERSION; }
const char* itcv_last_error(void) { return itcv::g_err; }

int itcv_profile_begin(void) {
  itcv::g_prof.clear();
  itcv::g_prof_on = true;
  return 0;
}
// stops recording, waits for the recorded events and returns the number of records
int itcv_profile_end(void) {
  itcv::g_prof_on = false;
  for (auto& r : itcv::g_prof) (void)hipEventSynchronize(r.stop);
  return (int)itcv::g_prof.size();
}
// record i: code = kind | KS<<4 | BM<<8 | up2<<16 | NS<<20 (kind 0 fwd fp32, 1 fwd split-bf16, 2 wgrad fp32,
// 3 wgrad split-bf16, 4 small-Cout direct, 5 small-Cin direct, 6 fwd on planes, 7 wgrad on planes and 8 band-form fwd on planes -- for 7 and 8 the KS field holds log2(W)), algorithmic FLOP,
// elapsed milliseconds
int itcv_profile_get(int i, int* code, double* flop, float* ms) {
  if (i < 0 || i >= (int)itcv::g_prof.size()) return itcv::fail("%s: index out of range", "itcv_profile_get");
  const auto& r = itcv::g_prof[i];
  *code = r.code;
  *flop = r.flop;
  if (hipEventElapsedTime(ms, r.start, r.stop) != hipSuccess) *ms = -1.f;
  return 0;
}
int itcv_profile_clear(void) {
  for (auto& r : itcv::g_prof) {
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  itcv::g_prof.clear();
  return 0;
}
}
