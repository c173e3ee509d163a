// vdl_jit.h -- run-time specialisation of the fused scans (vdl_jit.cpp)
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <vector>

#include "vdl_scan_desc.h"

namespace vdl {
namespace jit {

struct Shape { int nc = 0, u = 0; bool vec = false, grouped = false, der = false;
               bool census = false; };     // census: a staged scan's late loads also count the 128-byte lines they ask for (measurement builds)

// what is specialised: an aggregate scan, or the two passes of the projection scan (fused front; dimension scans are the
// select pass with bitmap_only set)
enum Kind : int { MSCAN = 0, SELECT = 1, TAKE = 2, FRONT = 3 };    // (TAKE: the take-side descriptor's fields; FRONT: the fused front in one pass)
const char *entry_name(Kind kind);
std::string entry_name(Kind kind, const MsArgs &C, const MScanDesc &D, const Shape &sh);
// the translation unit: the embedded device code + this scan's descriptor as constants + the kernel
std::string scan_source(Kind kind, const MsArgs &C, const MScanDesc &D, const Shape &sh);
std::string mscan_source(const MsArgs &C, const MScanDesc &D, const Shape &sh);
// the one-pass front: select-side args / descriptor (sh.nc columns), take-side ones (nct columns)
std::string front_source(const MsArgs &Cs, const MScanDesc &Ds, const MsArgs &Ct, const MScanDesc &Dt, const Shape &sh, int nct);
// hiprtc (no GPU needed); cached per process and under $VDL_JIT_CACHE.  false: `log` says why
bool compile(const std::string &src, const std::string &arch, std::vector<char> &code, std::string &log);

struct Kernel {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    ~Kernel();
};
std::shared_ptr<Kernel> load(const std::vector<char> &code, std::string &why, const std::string &entry);

}  // namespace jit
}  // namespace vdl
