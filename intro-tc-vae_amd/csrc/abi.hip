// Library-level entry points of libitcv_hip.so (error reporting, ABI version).
#include "conv_shared.h"

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

static thread_local int g_prof_depth = 0;   // nested scopes (an entry point calling another one) belong to the outermost
ProfScope::ProfScope(hipStream_t stream, int kind, int ks, int bm, int up2, int ns, double flop) : st(stream), slot(-1) {
  if (++g_prof_depth > 1 || !g_prof_on) return;
  ProfRec r;
  r.code = kind | (ks << 4) | (bm << 8) | (up2 << 16) | (ns << 20);
  r.flop = flop;
  if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
  g_prof.push_back(r);
  slot = (int)g_prof.size() - 1;
  g_prof_start = r.start, g_prof_stop = r.stop;   // consumed by the scope's first launch_timed()
}
ProfScope::~ProfScope() {
  if (--g_prof_depth > 0) return;
  if (slot >= 0 && g_prof_start) {   // nothing was launched through launch_timed(): drop the record
    (void)hipEventDestroy(g_prof[slot].start);
    (void)hipEventDestroy(g_prof[slot].stop);
    g_prof.pop_back();
  }
  g_prof_start = g_prof_stop = nullptr;
}
thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;
Options g_opt;
}

extern "C" {
int itcv_abi_version(void) { return ITCV_ABI_VERSION; }

int itcv_set_option(const char* name, int value) {
  if (!name) return itcv::fail("%s: null option name", "itcv_set_option");
  if (!strcmp(name, "band_m16")) {
    if (value != 0 && value != 1) return itcv::fail("%s: band_m16 takes 0 or 1 (got %lld)", "itcv_set_option", value);
    itcv::g_opt.band_m16 = value;
    return 0;
  }
  if (!strcmp(name, "band_persist_blocks")) {
    if (value < 0 || value > 1024) return itcv::fail("%s: band_persist_blocks takes 0..1024 (got %lld)", "itcv_set_option", value);
    itcv::g_opt.band_persist_blocks = value;
    return 0;
  }
  if (!strcmp(name, "wgrad_m16")) {
    if (value != 0 && value != 1) return itcv::fail("%s: wgrad_m16 takes 0 or 1 (got %lld)", "itcv_set_option", value);
    itcv::g_opt.wgrad_m16 = value;
    return 0;
  }
  if (!strcmp(name, "planes_mfma_waves")) {
    if (value != 4 && value != 8) return itcv::fail("%s: planes_mfma_waves takes 4 or 8 (got %lld)", "itcv_set_option", value);
    itcv::g_opt.planes_mfma_waves = value;
    return 0;
  }
  return itcv::fail("%s: unknown option", "itcv_set_option");
}
int itcv_get_option(const char* name) {
  if (name && !strcmp(name, "band_m16")) return itcv::g_opt.band_m16;
  if (name && !strcmp(name, "band_persist_blocks")) return itcv::g_opt.band_persist_blocks;
  if (name && !strcmp(name, "wgrad_m16")) return itcv::g_opt.wgrad_m16;
  if (name && !strcmp(name, "planes_mfma_waves")) return itcv::g_opt.planes_mfma_waves;
  return -1;
}
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
