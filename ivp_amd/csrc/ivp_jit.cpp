// ivp_jit.cpp -- user-defined right-hand sides compiled at run time with hiprtc.
//
// The reference lets a user implement `trait IVP { fn ode(&self, x, y, dydx) }` (src/ivp.rs:27-29)
// in host Rust.  On the GPU the right-hand side has to be device code, so the analogue is a HIP
// source snippet defining
//     __device__ void ode(double x, const double* y, double* dydx, const double* p);
// which is spliced in front of the very same kernel templates the built-in functors use
// (ivp_kargs.h + rk_core.h + rk_global.h are embedded in the library as text) and compiled for the
// context's gfx target.  One module per (method, output mode, fp mode) is built on first use.
#include "ivp_jit.h"

#include <hip/hiprtc.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/ivp_hip.h"
#include "rk_launch.h"
#include "ivp_jit_sources.inc"

namespace {

struct JitModule {
    hipModule_t mod = nullptr;
    hipFunction_t init = nullptr, chunk = nullptr, coop = nullptr;   // coop: lane-cooperative chunk kernel (rk_coop.h), optional
    hipFunction_t events = nullptr;   // deferred event refinement (rk_global.h event_kernel_body): full modules of problems with event functions
};

struct JitRhs {
    int device, n, np, ne;
    bool has_jac = false;   // the snippet defines jac(): IVP::jac override (src/ivp.rs:67-107)
    std::string ode_source;
    std::string arch;
    std::mutex mu;
    // (device, method, fp_mode, full, ctl): hipModuleLoadData binds a module to the device that is current when it is
    // loaded, so a handle shared by contexts on several GPUs keeps one module per device
    std::map<std::tuple<int, int, int, int, bool, bool>, JitModule> modules;   // ... , lane-cooperative module
    std::string log;
};

std::string join(const char *const *parts)
{
    std::string s;
    for (; *parts; ++parts) s += *parts;
    return s;
}

std::string build_source(const JitRhs &r, int method, int full_in, bool ctl, bool coop_only)
{
    // kernel flavour (rk_launch.h): 2 = log-only exists for the adaptive explicit methods of problems without event functions
    const int flavour = (full_in == 2 && r.ne == 0 && (method == IVP_RK23 || method == IVP_DOPRI5 || method == IVP_DOP853)) ? 2 : (full_in ? 1 : 0);
    const char *full = flavour == 2 ? "2" : (flavour ? "1" : "0");
    std::string s;
    s += "typedef unsigned int uint32_t;\ntypedef int int32_t;\ntypedef unsigned long long uint64_t;\ntypedef long long int64_t;\n";
    s += "#define IVP_HD __device__ __forceinline__\n";
    s += "#define IVP_NS ivp_jit\n";
    s += "#define IVP_USER_NE " + std::to_string(r.ne) + "\n";
    s += std::string("#define IVP_USER_JAC ") + (r.has_jac ? "1" : "0") + "\n";
    const bool group = r.n > IVP_MAX_N;   // wave-per-trajectory kernels (rk_group.h): user code defines ode_comp()
    if (group || coop_only) s += "#define IVP_HOIST 2\n";   // a wave that owns its SIMD: coefficients pinned in registers (rk_core.h KC)
    s += join(k_src_ivp_kargs_h);
    s += "\n// ---- user right-hand side ----\n";
    s += r.ode_source;
    s += "\n// ---- integrator ----\n";
    s += join(k_src_rk_core_h);
    char buf[4096];
    if (group) {
        s += join(k_src_bdf_core_h);
        s += join(k_src_rk_global_h);   // compact_append
        s += join(k_src_rk_group_h);
        s += join(k_src_bdf_group_h);
        std::snprintf(buf, sizeof buf,
                      "namespace ivp_jit { struct RhsUser { enum { N = %d, P = %d, NE = IVP_USER_NE };\n"
                      "  static __device__ __forceinline__ double ode_comp(int i, double x, const double* y, const double* p) { return ::ode_comp(i, x, y, p); }\n"
                      "#if IVP_USER_NE > 0\n"
                      "  static __device__ __forceinline__ void events(double x, const double* y, double* g, const double* p) { ::events(x, y, g, p); }\n"
                      "#endif\n"
                      "#if IVP_USER_JAC\n"
                      "  static __device__ __forceinline__ void jac_col(int col, double x, const double* y, double* column, const double* p) { ::jac_col(col, x, y, column, p); }\n"
                      "#endif\n"
                      "}; }\n"
                      "extern \"C\" __global__ __launch_bounds__(IVP_WAVE) void ivp_jit_init(const IvpKArgs a) { ivp_jit::group_init_body<%d, ivp_jit::RhsUser, %s, %d>(a); }\n"
                      "extern \"C\" __global__ __launch_bounds__(IVP_WAVE) void ivp_jit_chunk(const IvpKArgs a) { ivp_jit::group_chunk_body<%d, ivp_jit::RhsUser, %s, %d>(a); }\n",
                      r.n, r.np, method, full, ivp_group_width(r.n), method, full, ivp_group_width(r.n));
        s += buf;
        return s;
    }
    s += join(k_src_bdf_core_h);
    s += join(k_src_rk_global_h);
    std::snprintf(buf, sizeof buf,
                  "namespace ivp_jit { struct RhsUser { enum { N = %d, P = %d, NE = IVP_USER_NE };\n"
                  "  static IVP_HD void ode(double x, const double* y, double* d, const double* p) { ::ode(x, y, d, p); }\n"
                  "#if IVP_USER_NE > 0\n"
                  "  static IVP_HD void events(double x, const double* y, double* g, const double* p) { ::events(x, y, g, p); }\n"
                  "#endif\n"
                  "#if IVP_USER_JAC\n"
                  "  static IVP_HD void jac(double x, const double* y, double (&j)[N][N], const double* p) { ::jac(x, y, &j[0][0], p); }\n"
                  "#endif\n"
                  "}; }\n", r.n, r.np);
    s += buf;
    if (coop_only) {   // eight lanes per trajectory for the tail of a batch: its own module, pinned coefficients
        s += join(k_src_rk_coop_h);
        std::snprintf(buf, sizeof buf,
                      "extern \"C\" __global__ __launch_bounds__(IVP_WAVE) void ivp_jit_coop(const IvpKArgs a)\n"
                      "{ ivp_jit::coop_chunk_body<%d, ivp_jit::RhsUser, %s>(a); }\n", method, full);
        s += buf;
        return s;
    }
    std::snprintf(buf, sizeof buf,
                  "extern \"C\" __global__ __launch_bounds__(IVP_WAVE) void ivp_jit_init(const IvpKArgs a)\n"
                  "{ const uint32_t i = blockIdx.x * IVP_WAVE + threadIdx.x; if (i < a.B) ivp_jit::any_init_body<%d, ivp_jit::RhsUser, %s>(a, i); }\n"
                  "extern \"C\" __global__ __launch_bounds__(IVP_WAVE, IVP_MIN_WAVES) void ivp_jit_chunk(const IvpKArgs a)\n"
                  "{ ivp_jit::chunk_kernel_body<%d, ivp_jit::RhsUser, %s, %s>(a); }\n",
                  method, full, method, full,
                  (ctl && (method == IVP_RK23 || method == IVP_DOPRI5 || method == IVP_DOP853)) ? "true" : "false");
    s += buf;
    if (r.ne > 0 && flavour == 1 && method != IVP_BDF) {
        std::snprintf(buf, sizeof buf,
                      "extern \"C\" __global__ __launch_bounds__(IVP_WAVE) void ivp_jit_events(const IvpKArgs a)\n"
                      "{ ivp_jit::event_kernel_body<%d, ivp_jit::RhsUser>(a); }\n", method);
        s += buf;
    }
    return s;
}

// Optional on-disk cache of compiled code objects (hiprtc takes 1-2 s per module): set IVP_JIT_CACHE_DIR to an
// existing directory.  Key = FNV-1a of the complete generated source, the compile options and the hiprtc version.
// The directory must be PRIVATE to the user and trusted: entries are GPU code objects that are loaded and run as they
// are found (the 64-bit key locates an entry, it does not authenticate it).
std::string cache_path(const std::string &src, const std::string &opts)
{
    const char *dir = std::getenv("IVP_JIT_CACHE_DIR");
    if (!dir || !*dir) return std::string();
    int major = 0, minor = 0;
    (void)hiprtcVersion(&major, &minor);
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](const std::string &t) { for (unsigned char ch : t) { h ^= ch; h *= 1099511628211ull; } };
    mix(src); mix(opts); mix(std::to_string(major) + "." + std::to_string(minor));
    char name[64];
    std::snprintf(name, sizeof name, "/ivp_jit_%016llx.hsaco", (unsigned long long)h);
    return std::string(dir) + name;
}

int load_module(JitRhs &r, const std::vector<char> &code, JitModule *out, bool coop_only)
{
    if (hipModuleLoadData(&out->mod, code.data()) != hipSuccess) { r.log = "hipModuleLoadData failed"; return IVP_ERR_HIP; }
    if (coop_only) {
        if (hipModuleGetFunction(&out->coop, out->mod, "ivp_jit_coop") != hipSuccess) { r.log = "kernel lookup failed"; return IVP_ERR_HIP; }
        return IVP_OK;
    }
    if (hipModuleGetFunction(&out->init, out->mod, "ivp_jit_init") != hipSuccess ||
        hipModuleGetFunction(&out->chunk, out->mod, "ivp_jit_chunk") != hipSuccess) {
        r.log = "kernel lookup failed";
        return IVP_ERR_HIP;
    }
    if (hipModuleGetFunction(&out->events, out->mod, "ivp_jit_events") != hipSuccess) { out->events = nullptr; (void)hipGetLastError(); }   // only full modules with event functions have it
    return IVP_OK;
}

int compile_module(JitRhs &r, int method, int fp_mode, int full, bool ctl, JitModule *out, bool coop_only = false)
{
    const std::string src = build_source(r, method, full, ctl, coop_only);
    const std::string opt_key = r.arch + (fp_mode == IVP_FP_FAST ? "|fast" : "|strict");
    const std::string cpath = cache_path(src, opt_key);
    if (!cpath.empty()) {
        std::ifstream in(cpath, std::ios::binary);
        if (in) {
            std::vector<char> code((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
            if (!code.empty()) {
                if (!out) return IVP_OK;
                if (load_module(r, code, out, coop_only) == IVP_OK) return IVP_OK;
                (void)hipGetLastError();   // unreadable cache entry: fall through and compile
            }
        }
    }
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "ivp_user_rhs.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        r.log = "hiprtcCreateProgram failed";
        return IVP_ERR_JIT;
    }
    const std::string arch = "--offload-arch=" + r.arch;
    std::vector<const char *> opts = {arch.c_str(), "-O3", "-std=c++17"};
    opts.push_back("-ffp-contract=off");   // both arithmetic modes: the FMA mode's fused operations are written out (IVP_MA)
    opts.push_back(fp_mode == IVP_FP_FAST ? "-DIVP_FAST=1" : "-DIVP_FAST=0");
    const hiprtcResult cr = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    std::string log(ls, '\0');
    if (ls) hiprtcGetProgramLog(prog, &log[0]);
    if (cr != HIPRTC_SUCCESS) {
        r.log = "hiprtc: " + log;
        hiprtcDestroyProgram(&prog);
        return IVP_ERR_JIT;
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    if (!cpath.empty()) {   // best effort: write to a temporary name, then rename (atomic on POSIX)
        const std::string tmp = cpath + ".tmp" + std::to_string((unsigned long long)(uintptr_t)&r);
        std::ofstream o(tmp, std::ios::binary);
        if (o) {
            o.write(code.data(), (std::streamsize)code.size());
            o.close();
            if (!o || std::rename(tmp.c_str(), cpath.c_str()) != 0) std::remove(tmp.c_str());
        }
    }
    if (!out) return IVP_OK;  // compile-only check
    return load_module(r, code, out, coop_only);
}

}  // namespace

int ivp_jit_n_events(void *handle) { return handle ? ((JitRhs *)handle)->ne : 0; }

int ivp_jit_compile(int device, const char *ode_source, int n, int n_params, int n_events, unsigned flags, void **handle, std::string *log)
{
    JitRhs *r = new JitRhs();
    r->device = device;
    r->has_jac = (flags & IVP_RHS_HAS_JAC) != 0;
    r->n = n;
    r->np = n_params;
    r->ne = n_events;
    r->ode_source = ode_source;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.gcnArchName[0]) {
        r->arch = prop.gcnArchName;
        const size_t colon = r->arch.find(':');  // "gfx950:sramecc+:xnack-" -> "gfx950"
        if (colon != std::string::npos) r->arch.resize(colon);
    } else {
        r->arch = "gfx950";
    }
    // compile the default configuration now so that syntax errors surface at ivp_rhs_compile() time
    int rc = compile_module(*r, IVP_DOPRI5, IVP_FP_STRICT, n_events > 0, false, nullptr);
    if (rc == IVP_OK && r->has_jac) rc = compile_module(*r, IVP_BDF, IVP_FP_STRICT, n_events > 0, false, nullptr);   // jac() is only instantiated by BDF
    if (rc != IVP_OK) {
        if (log) *log = r->log;
        delete r;
        return rc;
    }
    *handle = r;
    return IVP_OK;
}

void ivp_jit_free(void *handle)
{
    JitRhs *r = (JitRhs *)handle;
    if (!r) return;
    for (auto &kv : r->modules)
        if (kv.second.mod) (void)hipModuleUnload(kv.second.mod);
    delete r;
}

const char *ivp_jit_last_log(void *handle) { return handle ? ((JitRhs *)handle)->log.c_str() : ""; }

void ivp_jit_dims(void *handle, int *n, int *np)
{
    JitRhs *r = (JitRhs *)handle;
    *n = r->n;
    *np = r->np;
}

hipError_t ivp_jit_launch(void *handle, int what, int method, int fp_mode, int full, const IvpKArgs &a, uint32_t lanes,
                          hipStream_t s)
{
    JitRhs *r = (JitRhs *)handle;
    JitModule m;
    {
        std::lock_guard<std::mutex> lk(r->mu);
        const bool ctl = a.has_ctl != 0;
        const bool coop = what == IVP_LAUNCH_COOP;   // the cooperative kernel reads the controller fields at run time anyway
        if (coop && !((r->ne == 0 || full) && (method == IVP_DOPRI5 || method == IVP_DOP853) && r->n <= IVP_MAX_N)) return hipErrorInvalidValue;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
        auto key = std::make_tuple(dev, method, fp_mode, full, coop ? false : ctl, coop);
        auto it = r->modules.find(key);
        if (it == r->modules.end()) {
            JitModule nm;
            if (compile_module(*r, method, fp_mode, full, coop ? false : ctl, &nm, coop) != IVP_OK) return hipErrorInvalidValue;   // r->log says why (ivp_jit_last_log)
            it = r->modules.emplace(key, nm).first;
        }
        m = it->second;
    }
    unsigned grid = (lanes + IVP_WAVE - 1) / IVP_WAVE;
    if (what == IVP_LAUNCH_CHUNK && a.lpw && r->n <= IVP_MAX_N) grid = (lanes + a.lpw - 1) / a.lpw;   // thin waves (ivp_kargs.h)
    if (r->n > IVP_MAX_N) {   // large n: a group of G lanes per trajectory, 64 / G trajectories per wave
        const unsigned per_wave = IVP_WAVE / (unsigned)ivp_group_width(r->n);
        grid = (lanes + per_wave - 1) / per_wave;
    }
    hipFunction_t fn = what == IVP_LAUNCH_INIT ? m.init : m.chunk;
    unsigned grid_y = 1;
    if (what == IVP_LAUNCH_EVENTS) {   // one lane per noted step: grid.y strides over a trajectory's noted steps
        if (!m.events) return hipErrorInvalidValue;
        fn = m.events;
        grid = (lanes + IVP_WAVE - 1) / IVP_WAVE;
        grid_y = 4;
    }
    if (what == IVP_LAUNCH_COOP) {
        if (!m.coop) return hipErrorInvalidValue;
        fn = m.coop;
        grid = (lanes + 7) / 8;
    }
    if (grid == 0) return hipSuccess;
    IvpKArgs ka = a;
    void *args[] = {&ka};
    return hipModuleLaunchKernel(fn, grid, grid_y, 1, IVP_WAVE, 1, 1, 0, s, args, nullptr);
}
