// vdl_jit.h -- run-time specialisation of the fused scans (vdl_jit.cpp)
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <vector>

#include "vdl_scan_desc.h"

namespace vdl {
namespace jit {

struct Shape { int nc = 0, u = 0; bool vec = false, grouped = false, der = false; };

// the translation unit: the embedded device code + this scan's descriptor as constants + the kernel `vdl_jit_mscan`
std::string mscan_source(const MsArgs &C, const MScanDesc &D, const Shape &sh);
// hiprtc (no GPU needed); cached per process and under $VDL_JIT_CACHE.  false: `log` says why
bool compile(const std::string &src, const std::string &arch, std::vector<char> &code, std::string &log);

struct Kernel {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    ~Kernel();
};
std::shared_ptr<Kernel> load(const std::vector<char> &code, std::string &why);

}  // namespace jit
}  // namespace vdl
