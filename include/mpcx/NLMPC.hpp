// mpc::NLMPC<> over the mpcx C ABI -- the non-linear front-end of libmpc++ (reference include/mpc/NLMPC.hpp) for the
// systems that exist as device functors in libmpcx.so.  C++20, header-only, needs only <mpcx.h> and the small matrix
// types of mpcx/LMPC.hpp.
//
// What differs from the reference, and why: its hooks are std::function closures called on the host solver thread
// (IDimensionable.hpp:94-149); a GPU kernel cannot call them.  Here the system is chosen with setModel(id, params)
// -- one of the reference's example systems compiled into the library (libmpc_amd/csrc/nlmpc_models.hpp, where a new
// one is a 20-line struct) -- and the closure setters throw, saying so.  Everything else keeps its name and meaning:
// setDiscretizationSamplingTime, setOptimizerParameters(NLParameters), setStateBounds / setInputBounds (vector + slice
// and matrix forms), optimize(x0, lastU), getLastResult, getOptimalSequence, plus optimizeBatch.
#pragma once

#include <functional>

#include "LMPC.hpp"

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
    std::vector<double> params_, zprev_;
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
    void need() const
    {
        if (!h_) throw std::runtime_error("NLMPC: call setModel() first (the system, cost and constraints are device functors)");
    }
    void build()
    {
        if (h_) { mpcx_nlmpc_destroy(h_); h_ = nullptr; }
        detail::check(mpcx_nlmpc_create(model_, ph_, ch_, ts_, params_.empty() ? nullptr : params_.data(), (int)params_.size(),
                                        device(), &h_), "mpcx_nlmpc_create");
        mpcx_nlmpc_dims d{};
        detail::check(mpcx_nlmpc_get_dims(h_, &d), "mpcx_nlmpc_get_dims");
        if (d.nx != nx_ || d.nu != nu_ || d.nineq != ineq_)
            throw std::runtime_error("NLMPC: the model's dimensions do not match the template arguments");
        push_parameters();
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
    [[noreturn]] static void closures()
    {
        throw std::runtime_error("this controller's system, objective and constraint functions are device functors selected with "
                                 "setModel(); host closures cannot run inside the kernel");
    }
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

    /// extension that replaces the closure setters: MPCX_MODEL_* and the constants its closures capture in the reference
    void setModel(int model_id, const std::vector<double> &params = {})
    {
        if (eq_ != 0) throw std::runtime_error("NLMPC: user equality constraints are not available");
        model_ = model_id; params_ = params;
        build();
    }
    bool setDiscretizationSamplingTime(const double ts)                         // NLMPC.hpp:80-90
    {
        ts_ = ts;
        if (h_) build();
        return true;
    }
    void setOptimizerParameters(const Parameters &param)                       // NLMPC.hpp:97-101
    {
        prm_ = dynamic_cast<const NLParameters &>(param);
        if (h_) push_parameters();
    }
    bool setLoggerLevel(Logger::LogLevel) { return true; }
    bool setLoggerPrefix(std::string) { return true; }
    void setInputScale(const cvec<Tnu>) { throw std::runtime_error("input scaling is not available on the device functors"); }
    void setStateScale(const cvec<Tnx>) { throw std::runtime_error("state scaling is not available on the device functors"); }
    template <class F> bool setStateSpaceFunction(F &&, float = 1e-10f) { closures(); }     // NLMPC.hpp:139-157
    template <class F> bool setOutputFunction(F &&) { closures(); }
    template <class F> bool setObjectiveFunction(F &&) { closures(); }
    template <class F> bool setIneqConFunction(F &&, float = 1e-10f) { closures(); }
    template <class F> bool setEqConFunction(F &&, float = 1e-10f) { closures(); }

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
    mpcx_nlmpc_t handle() { return h_; }
};

}  // namespace mpc
