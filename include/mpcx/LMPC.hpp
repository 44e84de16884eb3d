// mpc::LMPC<> over the mpcx C ABI -- C++20, header-only, no third-party dependency.
//
// Keeps the template API of libmpc++'s linear front-end (reference include/mpc/LMPC.hpp) for the
// one path this repository replaces, so that code written against the reference -- e.g. its
// examples/quadrotor_ex.cpp or test/LMPC/test_common.cpp:89-237 -- compiles against this header
// and runs its optimize() on an MI355X.  The reference gets its matrix types from Eigen, which
// is not part of this repository: mpc::mat / mpc::cvec (mpcx/matrix.hpp) are a deliberately small
// stand-in covering what controller set-up code and hook bodies use.  Storage is column-major
// doubles like Eigen's default (reference include/mpc/Types.hpp:42), which is what the C ABI
// expects, so nothing is converted on the way down.
//
// Every method forwards to the entry point of include/mpcx.h that replaces the corresponding
// reference call.  Added on top of the reference API:
// optimizeBatch() for B instances of this controller in one launch.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <initializer_list>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "../mpcx.h"
#include "matrix.hpp"      // mpc::mat / mpc::cvec / mpc::rvec

namespace mpc {

constexpr double inf = std::numeric_limits<double>::infinity();


// ---------------------------------------------------------------------------------------------
// types of the reference API (include/mpc/Types.hpp)
// ---------------------------------------------------------------------------------------------
struct HorizonSlice {                       // Types.hpp:57-82
    int start, end;
    HorizonSlice(int s, int e) : start(s), end(e) {}
    static HorizonSlice all() { return HorizonSlice{-1, -1}; }
};
enum ResultStatus { SUCCESS, MAX_ITERATION, INFEASIBLE, ERROR, UNKNOWN };      // Types.hpp:87-94

struct Parameters {                         // Types.hpp:99-115
protected:
    Parameters() = default;

public:
    virtual ~Parameters() = default;
    int maximum_iteration = 100;
    double time_limit = 0;
    bool enable_warm_start = false;
};
struct LParameters : Parameters {           // Types.hpp:146-161
    LParameters() = default;
    double alpha = 1.6;
    double rho = 1e-6;
    double eps_rel = 1e-4;
    double eps_abs = 1e-4;
    double eps_prim_inf = 1e-3;
    double eps_dual_inf = 1e-3;
    bool verbose = false;
    bool adaptive_rho = true;
    bool polish = true;
};

template <int Tnu = Dynamic>
struct Result {                             // Types.hpp:168-182
    int solver_status = 0;
    bool is_feasible = false;
    std::string solver_status_msg;
    double cost = 0;
    ResultStatus status = ResultStatus::UNKNOWN;
    cvec<Tnu> cmd;
};
template <int Tnx = Dynamic, int Tny = Dynamic, int Tnu = Dynamic, int Tph = Dynamic>
struct OptSequence {                        // Types.hpp:184-198: row i = horizon step i
    mat<Tph, Tnx> state;
    mat<Tph, Tny> output;
    mat<Tph, Tnu> input;
};

struct Logger {                             // level setters are accepted; this path does not log
    enum class LogLevel { DEEP, NORMAL, ALERT, NONE };
    using log_level = LogLevel;
};

struct SolutionStats {                      // what IMPC::getExecutionStats exposes (Profiler.hpp:199-229), host-side
    int numberOfSolutions = 0;
    double totalTimeSec = 0, minSec = 0, maxSec = 0;
    friend std::ostream &operator<<(std::ostream &os, const SolutionStats &s)
    {
        return os << "solutions: " << s.numberOfSolutions << " total (s): " << s.totalTimeSec << " min (s): " << s.minSec
                  << " max (s): " << s.maxSec << std::endl;
    }
};

/// Result<nu> for B instances, host-side structure of arrays (instance-major)
struct BatchResult {
    int batch = 0, nu = 0;
    std::vector<double> cmd, cost;
    std::vector<int32_t> status, solver_status, is_feasible;
};

// ---------------------------------------------------------------------------------------------
// mpc::LMPC
// ---------------------------------------------------------------------------------------------
namespace detail {
inline constexpr int dimp1(int ph) { return ph < 0 ? Dynamic : ph + 1; }
inline void check(int rc, const char *what)
{
    if (rc != MPCX_OK) throw std::runtime_error(std::string(what) + ": " + mpcx_last_error());
}
}  // namespace detail

template <int Tnx = Dynamic, int Tnu = Dynamic, int Tndu = Dynamic, int Tny = Dynamic, int Tph = Dynamic, int Tch = Dynamic>
class LMPC {
    mpcx_dims d_{};
    mpcx_lmpc_t h_ = nullptr;
    Result<Tnu> last_;
    OptSequence<Tnx, Tny, Tnu, detail::dimp1(Tph)> seq_;
    SolutionStats stats_;

    void create()
    {
        // MPCX_DEVICE=-1 builds a host-only handle (setters work, any solve throws): used by CPU-only tests
        const char *e = std::getenv("MPCX_DEVICE");
        detail::check(mpcx_lmpc_create(&d_, e ? std::atoi(e) : 0, &h_), "mpcx_lmpc_create");
        last_.cmd.resize(d_.nu, 1);
        seq_.state.resize(d_.ph + 1, d_.nx); seq_.output.resize(d_.ph + 1, d_.ny); seq_.input.resize(d_.ph + 1, d_.nu);
    }
    bool ok(int rc) const
    {
        if (rc == MPCX_OK) return true;
        if (rc == MPCX_E_INVALID) return false;      // the reference's setters return false here
        throw std::runtime_error(mpcx_last_error());
    }

public:
    LMPC() requires(Tnx >= 0 && Tnu >= 0 && Tndu >= 0 && Tny >= 0 && Tph >= 0 && Tch >= 0)
    {
        d_ = mpcx_dims{Tnx, Tnu, Tndu, Tny, Tph, Tch};
        create();
    }
    LMPC(const int &nx, const int &nu, const int &ndu, const int &ny, const int &ph, const int &ch)
    {
        d_ = mpcx_dims{nx, nu, ndu, ny, ph, ch};
        create();
    }
    LMPC(const LMPC &) = delete;
    LMPC &operator=(const LMPC &) = delete;
    ~LMPC() { mpcx_lmpc_destroy(h_); }

    // ---- not available on the linear front-end (reference LMPC.hpp:68-100) -----------------------
    bool setDiscretizationSamplingTime(const double) { throw std::runtime_error("Linear MPC supports only discrete time systems"); }
    void setInputScale(const cvec<Tnu>) { throw std::runtime_error("Linear MPC does not support input scaling"); }
    void setStateScale(const cvec<Tnx>) { throw std::runtime_error("Linear MPC does not support state scaling"); }
    bool setLoggerLevel(Logger::LogLevel) { return true; }
    bool setLoggerPrefix(std::string) { return true; }

    // ---- set-up ---------------------------------------------------------------------------------------
    void setOptimizerParameters(const Parameters &param)                     // LMPC.hpp:79
    {
        const auto &lp = dynamic_cast<const LParameters &>(param);
        mpcx_lparams q{lp.maximum_iteration, lp.time_limit, lp.enable_warm_start ? 1 : 0, lp.alpha, lp.rho, lp.eps_rel,
                       lp.eps_abs, lp.eps_prim_inf, lp.eps_dual_inf, lp.verbose ? 1 : 0, lp.adaptive_rho ? 1 : 0, lp.polish ? 1 : 0};
        detail::check(mpcx_lmpc_set_optimizer_parameters(h_, &q), "setOptimizerParameters");
    }
    bool setStateSpaceModel(const mat<Tnx, Tnx> &A, const mat<Tnx, Tnu> &B, const mat<Tny, Tnx> &C)   // LMPC.hpp:493
    {
        return ok(mpcx_lmpc_set_state_space_model(h_, A.data(), B.data(), C.data()));
    }
    bool setDisturbances(const mat<Tnx, Tndu> &Bd, const mat<Tny, Tndu> &Dd)                           // LMPC.hpp:518
    {
        return ok(mpcx_lmpc_set_disturbances(h_, Bd.data(), Dd.data()));
    }
    bool setObjectiveWeights(const mat<Tny, Tph> &OW, const mat<Tnu, Tph> &UW, const mat<Tnu, Tph> &DUW)   // LMPC.hpp:306
    {
        return ok(mpcx_lmpc_set_objective_weights(h_, OW.data(), UW.data(), DUW.data()));
    }
    bool setObjectiveWeights(const cvec<Tny> &ow, const cvec<Tnu> &uw, const cvec<Tnu> &duw, const HorizonSlice &s)   // LMPC.hpp:436
    {
        return ok(mpcx_lmpc_set_objective_weights_slice(h_, ow.data(), uw.data(), duw.data(), s.start, s.end));
    }
    bool setStateBounds(const mat<Tnx, Tph> &lo, const mat<Tnx, Tph> &hi) { return ok(mpcx_lmpc_set_state_bounds(h_, lo.data(), hi.data())); }
    bool setInputBounds(const mat<Tnu, Tch> &lo, const mat<Tnu, Tch> &hi) { return ok(mpcx_lmpc_set_input_bounds(h_, lo.data(), hi.data())); }
    bool setOutputBounds(const mat<Tny, Tph> &lo, const mat<Tny, Tph> &hi) { return ok(mpcx_lmpc_set_output_bounds(h_, lo.data(), hi.data())); }
    bool setStateBounds(const cvec<Tnx> &lo, const cvec<Tnx> &hi, const HorizonSlice &s)
    {
        return ok(mpcx_lmpc_set_state_bounds_slice(h_, lo.data(), hi.data(), s.start, s.end));
    }
    bool setInputBounds(const cvec<Tnu> &lo, const cvec<Tnu> &hi, const HorizonSlice &s)
    {
        return ok(mpcx_lmpc_set_input_bounds_slice(h_, lo.data(), hi.data(), s.start, s.end));
    }
    bool setOutputBounds(const cvec<Tny> &lo, const cvec<Tny> &hi, const HorizonSlice &s)
    {
        return ok(mpcx_lmpc_set_output_bounds_slice(h_, lo.data(), hi.data(), s.start, s.end));
    }
    bool setScalarConstraint(const double mn, const double mx, const cvec<Tnx> X, const cvec<Tnu> U, const HorizonSlice &s)   // LMPC.hpp:355
    {
        return ok(mpcx_lmpc_set_scalar_constraint_slice(h_, mn, mx, X.data(), U.data(), s.start, s.end));
    }
    bool setScalarConstraint(const unsigned int index, const double mn, const double mx, const cvec<Tnx> X, const cvec<Tnu> U)   // LMPC.hpp:409
    {
        return ok(mpcx_lmpc_set_scalar_constraint_index(h_, (int)index, mn, mx, X.data(), U.data()));
    }
    bool setReferences(const mat<Tny, Tph> y, const mat<Tnu, Tph> u, const mat<Tnu, Tph> du)           // LMPC.hpp:596
    {
        return ok(mpcx_lmpc_set_references(h_, y.data(), u.data(), du.data()));
    }
    bool setReferences(const cvec<Tny> y, const cvec<Tnu> u, const cvec<Tnu> du, const HorizonSlice &s)   // LMPC.hpp:616
    {
        return ok(mpcx_lmpc_set_references_slice(h_, y.data(), u.data(), du.data(), s.start, s.end));
    }
    bool setExogenousInputs(const mat<Tndu, Tph> &d) { return ok(mpcx_lmpc_set_exogenous_inputs(h_, d.data())); }   // LMPC.hpp:534
    bool setExogenousInputs(const cvec<Tndu> &d, const HorizonSlice &s)                                            // LMPC.hpp:550
    {
        return ok(mpcx_lmpc_set_exogenous_inputs_slice(h_, d.data(), s.start, s.end));
    }

    // ---- IMPC::optimize (reference IMPC.hpp:149-166), one instance through the batched kernels ----------
    Result<Tnu> optimize(const cvec<Tnx> x0, const cvec<Tnu> lastU)
    {
        const int n1 = d_.ph + 1;
        std::vector<double> ss((size_t)n1 * d_.nx), so((size_t)n1 * d_.ny), si((size_t)n1 * d_.nu);
        Result<Tnu> r;
        r.cmd.resize(d_.nu, 1);
        int32_t st = 4, sst = 0, feas = 0;
        const int rc = mpcx_lmpc_solve_host(h_, 1, x0.data(), lastU.data(), r.cmd.data(), &r.cost, &st, &sst, &feas,
                                            ss.data(), so.data(), si.data());
        if (rc != MPCX_OK) throw std::runtime_error(std::string("optimize: ") + mpcx_last_error());
        r.status = static_cast<ResultStatus>(st); r.solver_status = sst; r.is_feasible = feas != 0;
        for (int i = 0; i < n1; ++i) {          // row-major from the device -> column-major matrices
            for (int j = 0; j < d_.nx; ++j) seq_.state(i, j) = ss[(size_t)i * d_.nx + j];
            for (int j = 0; j < d_.ny; ++j) seq_.output(i, j) = so[(size_t)i * d_.ny + j];
            for (int j = 0; j < d_.nu; ++j) seq_.input(i, j) = si[(size_t)i * d_.nu + j];
        }
        last_ = r;
        stats_.numberOfSolutions++;
        return r;
    }
    Result<Tnu> getLastResult() { return last_; }
    OptSequence<Tnx, Tny, Tnu, detail::dimp1(Tph)> getOptimalSequence() { return seq_; }
    const SolutionStats &getExecutionStats() { return stats_; }
    void resetStats() { stats_ = SolutionStats{}; }

    // ---- extension: B instances of this controller in one launch (host arrays, instance-major) ----------
    BatchResult optimizeBatch(int batch, const double *x0, const double *lastU)
    {
        BatchResult R;
        R.batch = batch; R.nu = d_.nu;
        R.cmd.resize((size_t)batch * d_.nu); R.cost.resize(batch);
        R.status.resize(batch); R.solver_status.resize(batch); R.is_feasible.resize(batch);
        detail::check(mpcx_lmpc_solve_host(h_, batch, x0, lastU, R.cmd.data(), R.cost.data(), R.status.data(),
                                           R.solver_status.data(), R.is_feasible.data(), nullptr, nullptr, nullptr),
                      "optimizeBatch");
        return R;
    }
    /// device-pointer form: fill an mpcx_lmpc_batch and launch asynchronously on a hipStream_t
    int optimizeBatch(const mpcx_lmpc_batch &b, void *stream) { return mpcx_lmpc_solve_batch(h_, &b, stream); }
    mpcx_lmpc_t handle() { return h_; }
};

}  // namespace mpc
