// mpcx_nlmpc_create_from_source: user hooks given as C++ source, compiled at run time (hipRTC) into the two kernels of
// the NLMPC engine.  The reference takes its hooks as host closures (NLMPC::setStateSpaceFunction & co., NLMPC.hpp:139-281);
// a host that cannot compile device code itself -- the Python front-end, a C host -- passes the BODIES of those lambdas
// instead.  The generated translation unit wraps each body in a functor with the reference's parameter list
// (IDimensionable.hpp:94-149), puts them in a mpcx::HookSet (every call inlined) and instantiates evaluate_body / sqp_body
// of include/mpcx/nlmpc_engine.hpp for mpcx::HookModel.  The engine headers travel inside libmpcx.so (build/engine_blob.inc).
// hipRTC is bound at run time (libhiprtc.so.7), like RCCL in comm_capi.cpp.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mpcx.h"
#include "mpcx/nlmpc_device.hpp"

namespace mpcx {
int capi_fail(int code, const std::string &msg);
void nlmpc_plan_host(NlmpcDev &m);
}
extern "C" __attribute__((visibility("hidden"))) int mpcx_nlmpc_create_hooked(const mpcx_nlmpc_custom *c, double Ts, int device, void *jit, mpcx_nlmpc_t *out);

namespace {

struct Header { const char *name; std::vector<unsigned char> text; };
const std::vector<Header> &engine_headers()
{
    static const std::vector<Header> h = {
#include "build/engine_blob.inc"
    };
    return h;
}

// hipRTC entry points with their own signatures (hiprtc.h); hiprtcProgram is an opaque pointer, hiprtcResult an int
struct Rtc {
    void *lib = nullptr;
    int (*CreateProgram)(void **, const char *, const char *, int, const char *const *, const char *const *) = nullptr;
    int (*CompileProgram)(void *, int, const char *const *) = nullptr;
    int (*GetProgramLogSize)(void *, size_t *) = nullptr;
    int (*GetProgramLog)(void *, char *) = nullptr;
    int (*GetCodeSize)(void *, size_t *) = nullptr;
    int (*GetCode)(void *, char *) = nullptr;
    int (*DestroyProgram)(void **) = nullptr;
    std::string error;
};
Rtc &rtc()
{
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so.7"};
        for (const char *n : names) {
            r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.error = std::string("hipRTC not found: ") + dlerror(); return; }
        auto sym = [&](const char *n) { void *p = dlsym(r.lib, n); if (!p) r.error = std::string("hipRTC lacks ") + n; return p; };
        r.CreateProgram = reinterpret_cast<decltype(r.CreateProgram)>(sym("hiprtcCreateProgram"));
        r.CompileProgram = reinterpret_cast<decltype(r.CompileProgram)>(sym("hiprtcCompileProgram"));
        r.GetProgramLogSize = reinterpret_cast<decltype(r.GetProgramLogSize)>(sym("hiprtcGetProgramLogSize"));
        r.GetProgramLog = reinterpret_cast<decltype(r.GetProgramLog)>(sym("hiprtcGetProgramLog"));
        r.GetCodeSize = reinterpret_cast<decltype(r.GetCodeSize)>(sym("hiprtcGetCodeSize"));
        r.GetCode = reinterpret_cast<decltype(r.GetCode)>(sym("hiprtcGetCode"));
        r.DestroyProgram = reinterpret_cast<decltype(r.DestroyProgram)>(sym("hiprtcDestroyProgram"));
    });
    return r;
}

struct JitModule {
    hipModule_t mod = nullptr;
    hipFunction_t k_eval = nullptr, k_sqp = nullptr;
    int device = 0;
};

int waves_per_block(const mpcx::NlmpcDev &m) { return mpcx::nlmpc_waves_per_block(m); }      // the engine's own plan

int jit_launch_eval(void *ctx, const void *devp, const void *batchp, void *stream)
{
    auto *j = static_cast<JitModule *>(ctx);
    const auto *m = static_cast<const mpcx::NlmpcDev *>(devp);
    const auto *b = static_cast<const mpcx::NlmpcBatchDev *>(batchp);
    const int wpb = waves_per_block(*m);
    if (wpb < 1) return -2;
    int blocks = (b->batch + wpb - 1) / wpb;
    if (blocks > 4096) blocks = 4096;
    void *args[] = {const_cast<mpcx::NlmpcDev *>(m), const_cast<mpcx::NlmpcBatchDev *>(b)};
    const hipError_t e = hipModuleLaunchKernel(j->k_eval, blocks, 1, 1, wpb * 64, 1, 1, (unsigned)((size_t)wpb * m->lds_per_wave * sizeof(double)),
                                               reinterpret_cast<hipStream_t>(stream), args, nullptr);
    return e == hipSuccess ? 0 : -3;
}
int jit_launch_solve(void *ctx, const void *devp, const void *solvep, void *stream)
{
    auto *j = static_cast<JitModule *>(ctx);
    const auto *m = static_cast<const mpcx::NlmpcDev *>(devp);
    const auto *b = static_cast<const mpcx::NlmpcSolveDev *>(solvep);
    const int wpb = waves_per_block(*m);
    if (wpb < 1) return -2;
    const int blocks = (b->batch + wpb - 1) / wpb;
    void *args[] = {const_cast<mpcx::NlmpcDev *>(m), const_cast<mpcx::NlmpcSolveDev *>(b)};
    const hipError_t e = hipModuleLaunchKernel(j->k_sqp, blocks, 1, 1, wpb * 64, 1, 1, (unsigned)((size_t)wpb * m->lds_per_wave * sizeof(double)),
                                               reinterpret_cast<hipStream_t>(stream), args, nullptr);
    return e == hipSuccess ? 0 : -3;
}

std::string generate(const mpcx_nlmpc_source &s)
{
    auto I = [](int v) { return std::to_string(v); };
    std::string t;
    t += "#include \"mpcx/nlmpc_hooks.hpp\"\n";
    t += "namespace user {\n";
    t += "constexpr int num_states = " + I(s.nx) + ", num_inputs = " + I(s.nu) + ", num_output = " + I(s.ny) + ", pred_hor = " + I(s.ph) +
         ", ctrl_hor = " + I(s.ch) + ", ineq_c = " + I(s.nineq) + ", eq_c = " + I(s.neq_user) + ";\n";
    if (s.preamble) { t += s.preamble; t += "\n"; }
    t += "struct StateFn { __device__ void operator()(mpc::cvec<num_states> &dx, const mpc::cvec<num_states> &x, const mpc::cvec<num_inputs> &u, "
         "const unsigned int &step) const {\n";
    t += s.state_fn; t += "\n} };\n";
    t += "struct ObjFn { __device__ double operator()(const mpc::mat<pred_hor + 1, num_states> &x, const mpc::mat<pred_hor + 1, num_output> &y, "
         "const mpc::mat<pred_hor + 1, num_inputs> &u, const double &e) const {\n";
    t += s.objective_fn; t += "\n} };\n";
    if (s.ineq_fn) {
        t += "struct IneqFn { __device__ void operator()(mpc::cvec<ineq_c> &in_con, const mpc::mat<pred_hor + 1, num_states> &x, "
             "const mpc::mat<pred_hor + 1, num_output> &y, const mpc::mat<pred_hor + 1, num_inputs> &u, const double &e) const {\n";
        t += s.ineq_fn; t += "\n} };\n";
    } else t += "using IneqFn = mpcx::NoHook;\n";
    if (s.eq_fn) {
        t += "struct EqFn { __device__ void operator()(mpc::cvec<eq_c> &eq_con, const mpc::mat<pred_hor + 1, num_states> &x, "
             "const mpc::mat<pred_hor + 1, num_inputs> &u) const {\n";
        t += s.eq_fn; t += "\n} };\n";
    } else t += "using EqFn = mpcx::NoHook;\n";
    if (s.output_fn) {
        t += "struct OutFn { __device__ void operator()(mpc::cvec<num_output> &y, const mpc::cvec<num_states> &x, const mpc::cvec<num_inputs> &u, "
             "const unsigned int &step) const {\n";
        t += s.output_fn; t += "\n} };\n";
    } else t += "using OutFn = mpcx::NoHook;\n";
    t += "}  // namespace user\n";
    t += "using Hooks = mpcx::HookSet<user::StateFn, user::ObjFn, user::IneqFn, user::EqFn, user::OutFn>;\n";
    t += "using Model = mpcx::HookModel<user::num_states, user::num_inputs, user::num_output, user::pred_hor, user::ctrl_hor, user::ineq_c, "
         "user::eq_c, Hooks>;\n";
    t += "extern \"C\" __global__ __launch_bounds__(256) void mpcx_jit_evaluate(const mpcx::NlmpcDev M, const mpcx::NlmpcBatchDev B) "
         "{ mpcx::engine::evaluate_body<Model>(M, B); }\n";
    t += "extern \"C\" __global__ __launch_bounds__(256, mpcx::engine::kSqpWavesPerSimd<Model>) void mpcx_jit_sqp(const mpcx::NlmpcDev M, const mpcx::NlmpcSolveDev S) "
         "{ mpcx::engine::sqp_body<Model, true>(M, S); }\n";
    return t;
}

}  // namespace

namespace mpcx {
void nlmpc_jit_release(void *jit)
{
    auto *j = static_cast<JitModule *>(jit);
    if (!j) return;
    if (j->mod) (void)hipModuleUnload(j->mod);
    delete j;
}
}  // namespace mpcx

extern "C" {

/* testing aid (not in mpcx.h): the translation unit mpcx_nlmpc_create_from_source would compile */
int mpcx_nlmpc_debug_generated_source(const mpcx_nlmpc_source *src, char *out, int cap)
{
    if (!src || !src->state_fn || !src->objective_fn) return mpcx::capi_fail(MPCX_E_INVALID, "state_fn and objective_fn are required");
    const std::string t = generate(*src);
    if (out && cap > 0) { std::snprintf(out, (size_t)cap, "%s", t.c_str()); }
    return (int)t.size();
}

/* Compiles the hooks for gfx950; code_out (may be NULL) receives the code object.  No GPU is needed for this step, which is
 * what the CPU-only tests exercise; mpcx_nlmpc_create_from_source loads the result.  Returns the code size or an error.  */
__attribute__((visibility("hidden"))) int mpcx_nlmpc_compile_source(const mpcx_nlmpc_source *src, std::vector<char> *code_out)      // internal (C++ signature)
{
    using mpcx::capi_fail;
    if (!src || !src->state_fn || !src->objective_fn) return capi_fail(MPCX_E_INVALID, "state_fn and objective_fn are required");
    if (src->nx < 1 || src->nu < 1 || src->ny < 0 || src->nineq < 0 || src->neq_user < 0 || src->ph < 1 || src->ch < 1 || src->ch > src->ph)
        return capi_fail(MPCX_E_INVALID, "bad dimensions");
    if ((src->nineq > 0) != (src->ineq_fn != nullptr)) return capi_fail(MPCX_E_INVALID, "ineq_fn goes with nineq > 0");
    if ((src->neq_user > 0) != (src->eq_fn != nullptr)) return capi_fail(MPCX_E_INVALID, "eq_fn goes with neq_user > 0");
    Rtc &r = rtc();
    if (!r.error.empty()) return capi_fail(MPCX_E_DEVICE, r.error);
    const std::string text = generate(*src);
    std::vector<const char *> hsrc, hname;
    for (const Header &h : engine_headers()) { hsrc.push_back(reinterpret_cast<const char *>(h.text.data())); hname.push_back(h.name); }
    void *prog = nullptr;
    if (r.CreateProgram(&prog, text.c_str(), "mpcx_user_hooks.hip", (int)hsrc.size(), hsrc.data(), hname.data()) != 0)
        return capi_fail(MPCX_E_DEVICE, "hiprtcCreateProgram failed");
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=fast"};
    const int rc = r.CompileProgram(prog, 4, opts);
    if (rc != 0) {
        size_t n = 0;
        std::string log;
        if (r.GetProgramLogSize(prog, &n) == 0 && n > 1) { log.resize(n); (void)r.GetProgramLog(prog, log.data()); }
        (void)r.DestroyProgram(&prog);
        return capi_fail(MPCX_E_INVALID, "the hook sources do not compile:\n" + log);
    }
    size_t n = 0;
    if (r.GetCodeSize(prog, &n) != 0 || n == 0) { (void)r.DestroyProgram(&prog); return capi_fail(MPCX_E_DEVICE, "hiprtcGetCodeSize failed"); }
    if (code_out) { code_out->resize(n); (void)r.GetCode(prog, code_out->data()); }
    (void)r.DestroyProgram(&prog);
    return (int)n;
}
int mpcx_nlmpc_debug_compile_source(const mpcx_nlmpc_source *src) { return mpcx_nlmpc_compile_source(src, nullptr); }

int mpcx_nlmpc_create_from_source(const mpcx_nlmpc_source *src, double Ts, int device, mpcx_nlmpc_t *out)
{
    using mpcx::capi_fail;
    if (!out) return capi_fail(MPCX_E_INVALID, "null output handle");
    if (hipSetDevice(device) != hipSuccess) return capi_fail(MPCX_E_DEVICE, "hipSetDevice failed: no usable HIP device");
    {
        // the engine is written for gfx950 and the hooks are compiled for it: say so instead of failing inside hipModuleLoadData
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
            return capi_fail(MPCX_E_UNSUPPORTED, std::string("run-time compiled hooks target gfx950, this device is ") + prop.gcnArchName);
    }
    std::vector<char> code;
    const int n = mpcx_nlmpc_compile_source(src, &code);
    if (n < 0) return n;
    auto *j = new JitModule;
    j->device = device;
    if (hipModuleLoadData(&j->mod, code.data()) != hipSuccess ||
        hipModuleGetFunction(&j->k_eval, j->mod, "mpcx_jit_evaluate") != hipSuccess ||
        hipModuleGetFunction(&j->k_sqp, j->mod, "mpcx_jit_sqp") != hipSuccess) {
        mpcx::nlmpc_jit_release(j);
        return capi_fail(MPCX_E_DEVICE, "could not load the compiled hooks");
    }
    mpcx_nlmpc_custom c{};
    c.nx = src->nx; c.nu = src->nu; c.ny = src->ny; c.ph = src->ph; c.ch = src->ch; c.nineq = src->nineq; c.neq_user = src->neq_user;
    c.has_output = src->output_fn ? 1 : 0; c.vector_hooks = 1;
    c.hooks = nullptr; c.hooks_bytes = 0;             // the generated functors capture nothing
    c.launch_evaluate = jit_launch_eval; c.launch_solve = jit_launch_solve; c.launch_ctx = j;
    return mpcx_nlmpc_create_hooked(&c, Ts, device, j, out);
}

}  // extern "C"
