// mpc::NLMPC<> over the mpcx C ABI -- the non-linear front-end of libmpc++ (reference include/mpc/NLMPC.hpp).
// C++20, header-only; needs <mpcx.h>, the small matrix types of mpcx/matrix.hpp and, for device hooks, hipcc.
//
// The reference's hooks are std::function closures called on the host solver thread (IDimensionable.hpp:94-149); a GPU
// kernel cannot call host code.  Three ways to give this controller its system, cost and constraints, all ending in the
// same engine (mpcx/nlmpc_engine.hpp):
//   1. setStateSpaceFunction / setOutputFunction / setObjectiveFunction / setIneqConFunction / setEqConFunction with the
//      reference's names, order and parameter lists, when this header is compiled by hipcc: the arguments are device
//      callables -- a lambda whose captures are by value and trivially copyable (`[=] __device__`, or no capture at all), or
//      a functor struct -- and their bodies are the reference's.  Set one at a time their types are never known together,
//      so each is reached through a device function pointer; setHooks(f, obj, ineq, eq, out) takes them together and has
//      every call inlined (same results, faster).
//   2. setHookSources(...): the lambda BODIES as text, compiled at run time (hipRTC) -- works from a host compiler too.
//   3. setModel(id, params): one of the reference's example systems built into libmpcx.so (the fastest form: component-
//      wise constraints with declared structure, mpcx/nlmpc_models.hpp).
// Everything else keeps its name and meaning: setDiscretizationSamplingTime, setOptimizerParameters(NLParameters),
// setInputScale / setStateScale, setStateBounds / setInputBounds (vector + slice and matrix forms), optimize(x0, lastU),
// getLastResult, getOptimalSequence, plus optimizeBatch.  The handle is (re)built lazily by the first optimize() after a
// change of hooks or sampling time.
#pragma once

#include <cstring>
#include <functional>
#include <memory>

#include "LMPC.hpp"
#if defined(__HIPCC__)
#include "nlmpc_hooks.hpp"
#endif

namespace mpc {

struct NLParameters : Parameters {           // Types.hpp:116-144
    double relative_ftol = -1, relative_xtol = -1, absolute_ftol = -1, absolute_xtol = -1;
    bool hard_constraints = true;
};

template <int Tnx = Dynamic, int Tnu = Dynamic, int Tny = Dynamic, int Tph = Dynamic, int Tch = Dynamic, int Tineq = Dynamic,
          int Teq = Dynamic>
class NLMPC {
    int nx_, nu_, ny_, ph_, ch_, ineq_, eq_;
    mpcx_nlmpc_t h_ = nullptr;
    int model_ = 0;
    double ts_ = 0.0;
    bool dirty_ = false;                       // hooks or sampling time changed: rebuild before the next solve
    std::vector<double> params_, zprev_, su_, ss_;
    // device hooks (mpcx/nlmpc_hooks.hpp): the closure objects as bytes and the launch thunks instantiated for their types
    std::vector<unsigned char> hook_blob_;
    int (*thunk_eval_)(void *, const void *, const void *, void *) = nullptr;
    int (*thunk_solve_)(void *, const void *, const void *, void *) = nullptr;
    bool hook_out_ = false, hooks_erased_ = false;
    // hook sources (hipRTC)
    struct Sources { bool set = false; std::string pre, f, obj, ineq, eq, out; bool has_ineq = false, has_eq = false, has_out = false; } src_;
    NLParameters prm_;
    bool have_prev_ = false;
    Result<Tnu> last_;
    OptSequence<Tnx, Tny, Tnu, detail::dimp1(Tph)> seq_;
    SolutionStats stats_;
    struct Bound { bool state; std::vector<double> lo, hi; int a, b; };
    std::vector<Bound> bounds_;

    static int device()
    {
        const char *e = std::getenv("MPCX_DEVICE");
        return e ? std::atoi(e) : 0;
    }
    void need()
    {
        if (dirty_ || !h_) build();
    }
    void build()
    {
        if (h_) { mpcx_nlmpc_destroy(h_); h_ = nullptr; }
        if (thunk_solve_) {
            mpcx_nlmpc_custom c{};
            c.nx = nx_; c.nu = nu_; c.ny = ny_; c.ph = ph_; c.ch = ch_; c.nineq = ineq_; c.neq_user = eq_;
            c.has_output = hook_out_ ? 1 : 0; c.vector_hooks = 1;
            c.hooks = hook_blob_.data(); c.hooks_bytes = (int)hook_blob_.size();
            c.launch_evaluate = thunk_eval_; c.launch_solve = thunk_solve_; c.launch_ctx = nullptr;
            detail::check(mpcx_nlmpc_create_custom(&c, ts_, device(), &h_), "mpcx_nlmpc_create_custom");
        } else if (src_.set) {
            mpcx_nlmpc_source q{nx_, nu_, ny_, ph_, ch_, ineq_, eq_, src_.pre.empty() ? nullptr : src_.pre.c_str(), src_.f.c_str(),
                                src_.obj.c_str(), src_.has_ineq ? src_.ineq.c_str() : nullptr, src_.has_eq ? src_.eq.c_str() : nullptr,
                                src_.has_out ? src_.out.c_str() : nullptr};
            detail::check(mpcx_nlmpc_create_from_source(&q, ts_, device(), &h_), "mpcx_nlmpc_create_from_source");
        } else if (model_ != 0) {
            detail::check(mpcx_nlmpc_create(model_, ph_, ch_, ts_, params_.empty() ? nullptr : params_.data(), (int)params_.size(),
                                            device(), &h_), "mpcx_nlmpc_create");
        } else {
            throw std::runtime_error("NLMPC: no system yet -- set the hooks (setStateSpaceFunction & co., setHooks, setHookSources) or "
                                     "pick a built-in model (setModel)");
        }
        dirty_ = false;
        mpcx_nlmpc_dims d{};
        detail::check(mpcx_nlmpc_get_dims(h_, &d), "mpcx_nlmpc_get_dims");
        if (d.nx != nx_ || d.nu != nu_ || d.nineq != ineq_ || d.neq_user != eq_)
            throw std::runtime_error("NLMPC: the model's dimensions do not match the template arguments");
        push_parameters();
        if (!su_.empty()) detail::check(mpcx_nlmpc_set_input_scale(h_, su_.data()), "setInputScale");
        if (!ss_.empty()) detail::check(mpcx_nlmpc_set_state_scale(h_, ss_.data()), "setStateScale");
        for (const auto &b : bounds_)
            (void)(b.state ? mpcx_nlmpc_set_state_bounds_slice(h_, b.lo.data(), b.hi.data(), b.a, b.b)
                           : mpcx_nlmpc_set_input_bounds_slice(h_, b.lo.data(), b.hi.data(), b.a, b.b));
        have_prev_ = false;
    }
    void push_parameters()
    {
        mpcx_nlparams q{prm_.maximum_iteration, prm_.time_limit, prm_.enable_warm_start ? 1 : 0, prm_.relative_ftol, prm_.relative_xtol,
                        prm_.absolute_ftol, prm_.absolute_xtol, prm_.hard_constraints ? 1 : 0};
        detail::check(mpcx_nlmpc_set_optimizer_parameters(h_, &q), "setOptimizerParameters");
    }
    [[noreturn]] static void host_compiler()
    {
        throw std::runtime_error("NLMPC: a closure can only become device code when this translation unit is compiled by hipcc; "
                                 "with a host compiler pass the hook bodies as text (setHookSources) or pick a built-in model (setModel)");
    }
#if defined(__HIPCC__)
    static constexpr bool kStatic = Tnx >= 0 && Tnu >= 0 && Tny >= 0 && Tph >= 0 && Tch >= 0 && Tineq >= 0 && Teq >= 0;
    using Erased = mpcx::ErasedHooks<(Tnx > 0 ? Tnx : 1), (Tnu > 0 ? Tnu : 1), (Tny > 0 ? Tny : 0), (Tph > 0 ? Tph : 1), (Tch > 0 ? Tch : 1),
                                     (Tineq > 0 ? Tineq : 0), (Teq > 0 ? Teq : 0)>;
    template <class Model> static int thunk_eval(void *ctx, const void *m, const void *b, void *s)
    {
        return mpcx::engine::launch_evaluate<Model>(ctx, static_cast<const mpcx::NlmpcDev *>(m), static_cast<const mpcx::NlmpcBatchDev *>(b), s);
    }
    template <class Model> static int thunk_solve(void *ctx, const void *m, const void *b, void *s)
    {
        return mpcx::engine::launch_solve<Model>(ctx, static_cast<const mpcx::NlmpcDev *>(m), static_cast<const mpcx::NlmpcSolveDev *>(b), s);
    }
    // one hook arrives: keep its closure bytes and the device address of its trampoline in the erased table
    template <class Fn> bool set_erased(Fn &&fill)
    {
        static_assert(kStatic, "device hooks need compile-time dimensions (the hook signatures carry them)");
        using Model = mpcx::HookModel<Tnx, Tnu, Tny, Tph, Tch, Tineq, Teq, Erased>;
        if (hipSetDevice(device()) != hipSuccess) throw std::runtime_error("NLMPC: no usable HIP device");
        // Hooks reached through function pointers run on the kernel's dynamic stack: the runtime sizes a lane's scratch as
        // max(the kernel's own frame, hipLimitStackSize), so the limit has to cover the kernel's frame (a few KiB of spills
        // and the matrix views handed to the hooks by reference) PLUS the frames of the hooks themselves.
        size_t stack = 0;
        const size_t want = 16384 + 4 * sizeof(mpc::mat<(Tph > 0 ? Tph : 1) + 1, (Tnx > Tnu ? (Tnx > Tny ? Tnx : Tny) : (Tnu > Tny ? Tnu : Tny))>);
        if (hipDeviceGetLimit(&stack, hipLimitStackSize) == hipSuccess && stack < want) (void)hipDeviceSetLimit(hipLimitStackSize, want);
        if (!hooks_erased_) { hook_blob_.assign(sizeof(Erased), 0); new (hook_blob_.data()) Erased(); hooks_erased_ = true; hook_out_ = false; }
        const bool ok = fill(*reinterpret_cast<Erased *>(hook_blob_.data()));
        thunk_eval_ = &thunk_eval<Model>; thunk_solve_ = &thunk_solve<Model>;
        src_.set = false; model_ = 0; dirty_ = true;
        return ok;
    }
#endif
    bool bound(bool state, const double *lo, const double *hi, int n, int a, int b)
    {
        bounds_.push_back(Bound{state, std::vector<double>(lo, lo + n), std::vector<double>(hi, hi + n), a, b});
        if (!h_) return true;
        const int rc = state ? mpcx_nlmpc_set_state_bounds_slice(h_, lo, hi, a, b) : mpcx_nlmpc_set_input_bounds_slice(h_, lo, hi, a, b);
        if (rc == MPCX_OK) return true;
        bounds_.pop_back();
        if (rc == MPCX_E_INVALID) return false;
        throw std::runtime_error(mpcx_last_error());
    }
    void init_outputs()
    {
        last_.cmd.resize(nu_, 1);
        seq_.state.resize(ph_ + 1, nx_); seq_.output.resize(ph_ + 1, ny_); seq_.input.resize(ph_ + 1, nu_);
    }

public:
    NLMPC() requires(Tnx >= 0 && Tnu >= 0 && Tny >= 0 && Tph >= 0 && Tch >= 0 && Tineq >= 0 && Teq >= 0)
        : nx_(Tnx), nu_(Tnu), ny_(Tny), ph_(Tph), ch_(Tch), ineq_(Tineq), eq_(Teq) { init_outputs(); }
    NLMPC(const int &nx, const int &nu, const int &ny, const int &ph, const int &ch, const int &ineq, const int &eq)
        : nx_(nx), nu_(nu), ny_(ny), ph_(ph), ch_(ch), ineq_(ineq), eq_(eq) { init_outputs(); }
    NLMPC(const NLMPC &) = delete;
    NLMPC &operator=(const NLMPC &) = delete;
    ~NLMPC() { if (h_) mpcx_nlmpc_destroy(h_); }

    /// extension: one of the systems built into the library, MPCX_MODEL_* and the constants its closures capture in the reference
    void setModel(int model_id, const std::vector<double> &params = {})
    {
        model_ = model_id; params_ = params;
        thunk_eval_ = nullptr; thunk_solve_ = nullptr; hooks_erased_ = false; src_.set = false;
        build();
    }
    /// extension: the hook bodies as C++ text, compiled at run time (parameter names and scope: mpcx_nlmpc_create_from_source)
    void setHookSources(const std::string &state_fn, const std::string &objective_fn, const std::string &ineq_fn = "",
                        const std::string &eq_fn = "", const std::string &output_fn = "", const std::string &preamble = "")
    {
        src_ = Sources{true, preamble, state_fn, objective_fn, ineq_fn, eq_fn, output_fn, !ineq_fn.empty(), !eq_fn.empty(), !output_fn.empty()};
        thunk_eval_ = nullptr; thunk_solve_ = nullptr; hooks_erased_ = false; model_ = 0; dirty_ = true;
    }
    bool setDiscretizationSamplingTime(const double ts)                         // NLMPC.hpp:80-90
    {
        ts_ = ts;
        dirty_ = true;
        return true;
    }
    void setOptimizerParameters(const Parameters &param)                       // NLMPC.hpp:97-101
    {
        prm_ = dynamic_cast<const NLParameters &>(param);
        if (h_) push_parameters();
    }
    bool setLoggerLevel(Logger::LogLevel) { return true; }
    bool setLoggerPrefix(std::string) { return true; }
    void setInputScale(const cvec<Tnu> scaling)                                // NLMPC.hpp:108 -> Mapping::setInputScaling
    {
        su_.assign(scaling.data(), scaling.data() + nu_);
        if (h_ && !dirty_) detail::check(mpcx_nlmpc_set_input_scale(h_, su_.data()), "setInputScale");
    }
    void setStateScale(const cvec<Tnx> scaling)                                // NLMPC.hpp:123 -> Mapping::setStateScaling
    {
        ss_.assign(scaling.data(), scaling.data() + nx_);
        if (h_ && !dirty_) detail::check(mpcx_nlmpc_set_state_scale(h_, ss_.data()), "setStateScale");
    }
#if defined(__HIPCC__)
    // The reference's closure setters (NLMPC.hpp:139-281).  The tolerance arguments are accepted for source compatibility;
    // feasibility is reported against the reference's defaults (1e-10, NLMPC.hpp:166,229,262).
    template <class F> bool setStateSpaceFunction(F handle, float = 1e-10f)                // NLMPC.hpp:165
    {
        return set_erased([&](Erased &e) { mpcx::hookdetail::stash(e.cdyn, handle); return mpcx::hookdetail::resolve<F>(e.pdyn); });
    }
    template <class F> bool setOutputFunction(F handle)                                    // NLMPC.hpp:202
    {
        const bool ok = set_erased([&](Erased &e) { mpcx::hookdetail::stash(e.cout_, handle); return mpcx::hookdetail::resolve<F>(e.pout); });
        hook_out_ = true;
        return ok;
    }
    template <class F> bool setObjectiveFunction(F handle)                                 // NLMPC.hpp:139
    {
        return set_erased([&](Erased &e) { mpcx::hookdetail::stash(e.cobj, handle); return mpcx::hookdetail::resolve<F>(e.pobj); });
    }
    template <class F> bool setIneqConFunction(F handle, float = 1e-10f)                   // NLMPC.hpp:228
    {
        return set_erased([&](Erased &e) { mpcx::hookdetail::stash(e.cineq, handle); return mpcx::hookdetail::resolve<F>(e.pineq); });
    }
    template <class F> bool setEqConFunction(F handle, float = 1e-10f)                     // NLMPC.hpp:261
    {
        return set_erased([&](Erased &e) { mpcx::hookdetail::stash(e.ceq, handle); return mpcx::hookdetail::resolve<F>(e.peq); });
    }
    /// extension: all hooks at once -- their types are known together, every call is inlined into the kernels.  Pass
    /// mpcx::NoHook{} for a hook the controller does not have.
    template <class FDyn, class FObj, class FIneq = mpcx::NoHook, class FEq = mpcx::NoHook, class FOut = mpcx::NoHook>
    bool setHooks(FDyn f, FObj obj, FIneq ineq = {}, FEq eq = {}, FOut out = {})
    {
        static_assert(kStatic, "device hooks need compile-time dimensions (the hook signatures carry them)");
        using Set = mpcx::HookSet<FDyn, FObj, FIneq, FEq, FOut>;
        using Model = mpcx::HookModel<Tnx, Tnu, Tny, Tph, Tch, Tineq, Teq, Set>;
        static_assert(__is_trivially_copyable(Set), "hooks must capture trivially copyable values (by value)");
        const Set set{f, obj, ineq, eq, out};
        hook_blob_.assign(sizeof(Set), 0);
        std::memcpy(hook_blob_.data(), &set, sizeof(Set));
        hooks_erased_ = false; hook_out_ = !__is_same(FOut, mpcx::NoHook);
        thunk_eval_ = &thunk_eval<Model>; thunk_solve_ = &thunk_solve<Model>;
        src_.set = false; model_ = 0; dirty_ = true;
        return true;
    }
#else
    template <class F> bool setStateSpaceFunction(F &&, float = 1e-10f) { host_compiler(); }     // NLMPC.hpp:139-281
    template <class F> bool setOutputFunction(F &&) { host_compiler(); }
    template <class F> bool setObjectiveFunction(F &&) { host_compiler(); }
    template <class F> bool setIneqConFunction(F &&, float = 1e-10f) { host_compiler(); }
    template <class F> bool setEqConFunction(F &&, float = 1e-10f) { host_compiler(); }
#endif

    bool setStateBounds(const cvec<Tnx> &lo, const cvec<Tnx> &hi, const HorizonSlice &s)    // NLMPC.hpp:346-358
    {
        return bound(true, lo.data(), hi.data(), nx_, s.start, s.end);
    }
    bool setInputBounds(const cvec<Tnu> &lo, const cvec<Tnu> &hi, const HorizonSlice &s)    // NLMPC.hpp:360-372
    {
        return bound(false, lo.data(), hi.data(), nu_, s.start, s.end);
    }
    bool setStateBounds(const mat<Tnx, Tph> &lo, const mat<Tnx, Tph> &hi)                   // NLMPC.hpp:285-299
    {
        bool res = true;
        for (int i = 0; i < ph_; ++i) res &= bound(true, lo.data() + (size_t)i * nx_, hi.data() + (size_t)i * nx_, nx_, i, i + 1);
        return res;
    }
    bool setInputBounds(const mat<Tnu, Tch> &lo, const mat<Tnu, Tch> &hi)                   // NLMPC.hpp:301-316
    {
        bool res = true;
        for (int i = 0; i < ch_; ++i) res &= bound(false, lo.data() + (size_t)i * nu_, hi.data() + (size_t)i * nu_, nu_, i, i + 1);
        return res;
    }
    template <class A, class B> bool setOutputBounds(const A &, const B &, const HorizonSlice & = HorizonSlice::all())
    {
        throw std::runtime_error("Output constraints cannot be set for this type of MPC");   // NLMPC.hpp:318-325
    }

    Result<Tnu> optimize(const cvec<Tnx> x0, const cvec<Tnu> lastU)                          // IMPC.hpp:149-166
    {
        need();
        mpcx_nlmpc_dims d{};
        detail::check(mpcx_nlmpc_get_dims(h_, &d), "mpcx_nlmpc_get_dims");
        const int n1 = ph_ + 1;
        std::vector<double> ss((size_t)n1 * nx_), si((size_t)n1 * nu_), z(d.nz);
        Result<Tnu> r;
        r.cmd.resize(nu_, 1);
        int32_t st = 4, sst = 0, feas = 0, it = 0;
        const bool warm = prm_.enable_warm_start && have_prev_;                              // NLOptimizer.hpp:431-510
        const int rc = mpcx_nlmpc_solve_host(h_, 1, x0.data(), lastU.data(), warm ? zprev_.data() : nullptr, r.cmd.data(), &r.cost, &st,
                                             &sst, &feas, &it, z.data(), ss.data(), si.data());
        if (rc != MPCX_OK) throw std::runtime_error(std::string("optimize: ") + mpcx_last_error());
        r.status = static_cast<ResultStatus>(st); r.solver_status = sst; r.is_feasible = feas != 0;
        if (r.status != ResultStatus::ERROR) { zprev_ = z; have_prev_ = true; }
        for (int i = 0; i < n1; ++i) {
            for (int j = 0; j < nx_; ++j) seq_.state(i, j) = ss[(size_t)i * nx_ + j];
            for (int j = 0; j < nu_; ++j) seq_.input(i, j) = si[(size_t)i * nu_ + j];
        }
        last_ = r;
        stats_.numberOfSolutions++;
        return r;
    }
    Result<Tnu> getLastResult() { return last_; }
    OptSequence<Tnx, Tny, Tnu, detail::dimp1(Tph)> getOptimalSequence() { return seq_; }
    const SolutionStats &getExecutionStats() { return stats_; }
    void resetStats() { stats_ = SolutionStats{}; }

    /// extension: B instances in one launch (host arrays, instance-major); z_warm may be null
    BatchResult optimizeBatch(int batch, const double *x0, const double *lastU, const double *z_warm = nullptr, double *z_out = nullptr)
    {
        need();
        BatchResult R;
        R.batch = batch; R.nu = nu_;
        R.cmd.resize((size_t)batch * nu_); R.cost.resize(batch);
        R.status.resize(batch); R.solver_status.resize(batch); R.is_feasible.resize(batch);
        detail::check(mpcx_nlmpc_solve_host(h_, batch, x0, lastU, z_warm, R.cmd.data(), R.cost.data(), R.status.data(),
                                            R.solver_status.data(), R.is_feasible.data(), nullptr, z_out, nullptr, nullptr),
                      "optimizeBatch");
        return R;
    }
    int optimizeBatch(const mpcx_nlmpc_batch &b, void *stream) { need(); return mpcx_nlmpc_solve_batch(h_, &b, stream); }
    mpcx_nlmpc_t handle() { need(); return h_; }
};

}  // namespace mpc
