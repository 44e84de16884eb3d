// The NLMPC counterpart of ioptimizer_backend_test.cpp: a backend of the SHAPE of the reference's non-linear optimizer
// (include/mpc/NLMPC/NLOptimizer.hpp:30-404 -- onInit(), setParameters(const Parameters &), setModel(...), bindObjective() / bindEq() /
// bindUserIneq() / bindUserEq(), setStateBounds() / setInputBounds(), run(x0, u0), the members `result` and `sequence` of IOptimizer.hpp:24-58)
// whose run() is the C ABI of include/mpcx.h.  The reference's headers cannot be included here (Eigen, NLopt); the interface is declared
// with this repository's matrix types, member for member.
//
// What a host std::function cannot do -- run inside a kernel -- the backend does not pretend to: the hooks reach it as what they are in
// the reference's sources, C++ text (the lambda bodies of NLMPC::setStateSpaceFunction & co., NLMPC.hpp:139-281), which
// mpcx_nlmpc_create_from_source compiles for the device; setModel() / the bind*() calls collect them exactly where NLOptimizer collects the
// Model / Objective / Constraints objects.
//
// Checked: (api, no GPU) the sources of examples/vanderpol_ex.cpp compile for gfx950 through the backend's own path and an invalid bounds
// slice is refused; (solve, GPU) run() returns the result and the sequences of this repository's mpc::NLMPC<> front-end with the built-in
// Van der Pol system to the finite-difference noise, the status mapping of NLOptimizer.hpp:729-750 holds on a maximum-iteration stop and on
// an infeasible bound set, and runBatch() solves 128 perturbed instances in one launch with row 0 equal to run().
#include <mpc/NLMPC.hpp>
#include <mpcx.h>

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static int failures = 0;
#define REQUIRE(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

namespace mpc {
class IOptimizerShape {                                           // IOptimizer.hpp:24-58, run-time sizes
public:
    virtual ~IOptimizerShape() {}
    void initialize(int nx, int nu, int ny, int ph, int ch, int ineq, int eq) { nx_ = nx; nu_ = nu; ny_ = ny; ph_ = ph; ch_ = ch; ineq_ = ineq; eq_ = eq; onInit(); }
    virtual void onInit() = 0;
    virtual void setParameters(const Parameters &param) = 0;
    virtual void run(const cvec<> &x0, const cvec<> &u0) = 0;
    Result<> result;
    OptSequence<> sequence;

protected:
    int nx_ = 0, nu_ = 0, ny_ = 0, ph_ = 0, ch_ = 0, ineq_ = 0, eq_ = 0;
};

class MpcxNLOptimizer : public IOptimizerShape {
    mpcx_nlmpc_t h_ = nullptr;
    int device_;
    double ts_ = 0.0;
    std::string f_, out_, obj_, ineq_src_, eq_src_;
    bool has_out_ = false, has_ineq_ = false, has_eq_ = false, bound_ = false;
    NLParameters prm_;
    std::vector<double> zprev_;
    bool have_prev_ = false;

    mpcx_nlmpc_source source() const
    {
        return mpcx_nlmpc_source{nx_, nu_, ny_, ph_, ch_, ineq_, eq_, nullptr, f_.c_str(), obj_.c_str(), has_ineq_ ? ineq_src_.c_str() : nullptr,
                                 has_eq_ ? eq_src_.c_str() : nullptr, has_out_ ? out_.c_str() : nullptr};
    }

public:
    explicit MpcxNLOptimizer(int device) : device_(device) {}
    ~MpcxNLOptimizer() override { if (h_) mpcx_nlmpc_destroy(h_); }
    void onInit() override                                        // NLOptimizer.hpp:62-97
    {
        result.cmd.resize(nu_, 1);
        sequence.state.resize(ph_ + 1, nx_); sequence.output.resize(ph_ + 1, ny_); sequence.input.resize(ph_ + 1, nu_);
    }
    // NLOptimizer::setModel (:100-112): the system (Model.hpp: state and output functions, sampling time)
    void setModel(const std::string &state_fn, const std::string &output_fn, double Ts) { f_ = state_fn; out_ = output_fn; has_out_ = !output_fn.empty(); ts_ = Ts; bound_ = false; }
    // NLOptimizer::bindObjective / bindUserIneq / bindUserEq (:204-344): what Objective / Constraints evaluate
    bool bindObjective(const std::string &objective_fn) { obj_ = objective_fn; bound_ = false; return !obj_.empty(); }
    bool bindUserIneq(const std::string &ineq_fn) { if (ineq_ <= 0) return false; ineq_src_ = ineq_fn; has_ineq_ = true; bound_ = false; return true; }
    bool bindUserEq(const std::string &eq_fn) { if (eq_ <= 0) return false; eq_src_ = eq_fn; has_eq_ = true; bound_ = false; return true; }
    // NLOptimizer::bindEq (:231-259): the dynamics equalities are the engine's own transcription (Constraints.hpp:490-905) -- nothing to hand over
    bool bindEq() { return !f_.empty(); }
    // compile-only check of what has been bound (no device needed)
    bool sourcesCompile() const { const mpcx_nlmpc_source s = source(); return mpcx_nlmpc_debug_compile_source(&s) > 0; }      // (the code object's size)
    // the handle exists from the first call that needs it (the reference builds its nlopt::opt in onInit; this one needs the hooks first)
    bool ensure()
    {
        if (bound_ && h_) return true;
        if (h_) { mpcx_nlmpc_destroy(h_); h_ = nullptr; }
        const mpcx_nlmpc_source s = source();
        if (mpcx_nlmpc_create_from_source(&s, ts_, device_, &h_) != MPCX_OK) { h_ = nullptr; return false; }
        bound_ = true;
        setParameters(prm_);
        return true;
    }
    void setParameters(const Parameters &p) override              // NLOptimizer.hpp:129-195
    {
        prm_ = dynamic_cast<const NLParameters &>(p);
        if (!h_) return;
        mpcx_nlparams q{prm_.maximum_iteration, prm_.time_limit, prm_.enable_warm_start, prm_.relative_ftol, prm_.relative_xtol, prm_.absolute_ftol,
                        prm_.absolute_xtol, prm_.hard_constraints};
        if (mpcx_nlmpc_set_optimizer_parameters(h_, &q) != MPCX_OK) throw std::runtime_error(mpcx_last_error());
    }
    bool setStateBounds(const cvec<> &lo, const cvec<> &hi, const HorizonSlice &s)     // :346-378
    {
        return ensure() && mpcx_nlmpc_set_state_bounds_slice(h_, lo.data(), hi.data(), s.start, s.end) == MPCX_OK;
    }
    bool setInputBounds(const cvec<> &lo, const cvec<> &hi, const HorizonSlice &s)     // :380-404
    {
        return ensure() && mpcx_nlmpc_set_input_bounds_slice(h_, lo.data(), hi.data(), s.start, s.end) == MPCX_OK;
    }
    void run(const cvec<> &x0, const cvec<> &u0) override         // NLOptimizer.hpp:412-638
    {
        const int n1 = ph_ + 1, nz = ph_ * nx_ + ch_ * nu_ + 1;
        std::vector<double> z(nz), ss((size_t)n1 * nx_), si((size_t)n1 * nu_);
        int32_t st = 3, sst = 0, feas = 0, it = 0;
        const bool warm = prm_.enable_warm_start && have_prev_;
        if (!ensure() || mpcx_nlmpc_solve_host(h_, 1, x0.data(), u0.data(), warm ? zprev_.data() : nullptr, result.cmd.data(), &result.cost, &st, &sst,
                                               &feas, &it, z.data(), ss.data(), si.data()) != MPCX_OK) {
            result.status = ResultStatus::ERROR;                   // as NLOptimizer::run when nlopt throws (:561-570)
            result.solver_status_msg = mpcx_last_error();
            return;
        }
        result.status = static_cast<ResultStatus>(st); result.solver_status = sst; result.is_feasible = feas != 0;
        if (result.status != ResultStatus::ERROR) { zprev_ = z; have_prev_ = true; }
        for (int i = 0; i < n1; ++i) {
            for (int j = 0; j < nx_; ++j) sequence.state(i, j) = ss[(size_t)i * nx_ + j];
            for (int j = 0; j < nu_; ++j) sequence.input(i, j) = si[(size_t)i * nu_ + j];
        }
    }
    int runBatch(const mpcx_nlmpc_batch &b, hipStream_t s) { return ensure() ? mpcx_nlmpc_solve_batch(h_, &b, s) : MPCX_E_INVALID; }
};
}  // namespace mpc

// examples/vanderpol_ex.cpp:38-39, 54, 62-64: the bodies of the three lambdas
static const char *kStateFn = "dx(0) = ((1.0 - (x(1) * x(1))) * x(0)) - x(1) + u(0); dx(1) = x(0);";
static const char *kObjFn = "return x.array().square().sum() + u.array().square().sum();";
static const char *kIneqFn = "for (int i = 0; i < ineq_c; i++) { in_con(i) = u(i, 0) - 0.5; }";

int main(int argc, char **argv)
{
    const bool solve = argc > 1 && std::strcmp(argv[1], "solve") == 0;
    const int nx = 2, nu = 1, ny = 2, ph = 10, ch = 5, ineq = ph + 1, eq = 0;
    mpc::MpcxNLOptimizer opt(0);
    opt.initialize(nx, nu, ny, ph, ch, ineq, eq);
    opt.setModel(kStateFn, "", 0.1);
    REQUIRE(opt.bindObjective(kObjFn));
    REQUIRE(opt.bindEq());
    REQUIRE(opt.bindUserIneq(kIneqFn));
    REQUIRE(!opt.bindUserEq(""));                                               // eq_c = 0: nothing to bind (NLOptimizer.hpp:306-344 returns false)
    mpc::NLParameters prm;
    prm.maximum_iteration = 200;
    opt.setParameters(prm);
    mpc::cvec<> x0(nx, 1), u0(nu, 1);
    x0(0) = 0.0; x0(1) = 1.0; u0.setZero();

    if (!solve) {
        REQUIRE(opt.sourcesCompile());                                          // hipRTC cross-compiles for gfx950 without a device
        mpc::MpcxNLOptimizer bad(0);
        bad.initialize(nx, nu, ny, ph, ch, ineq, eq);
        bad.setModel("dx(0) = undeclared_symbol;", "", 0.1);
        bad.bindObjective(kObjFn); bad.bindUserIneq(kIneqFn);
        REQUIRE(!bad.sourcesCompile());
        REQUIRE(std::strlen(mpcx_last_error()) > 0);                            // the compiler's diagnostics
    } else {
        opt.run(x0, u0);
        REQUIRE(opt.result.status == mpc::ResultStatus::SUCCESS && opt.result.is_feasible);
        // the built-in system through this repository's mpc::NLMPC<> front-end
        mpc::NLMPC<> ref(nx, nu, ny, ph, ch, ineq, eq);
        ref.setDiscretizationSamplingTime(0.1);
        ref.setOptimizerParameters(prm);
        ref.setModel(MPCX_MODEL_VANDERPOL);
        const auto r = ref.optimize(x0, u0);
        const auto seq = ref.getOptimalSequence();
        REQUIRE(r.status == opt.result.status);
        REQUIRE(std::fabs(r.cmd(0) - opt.result.cmd(0)) <= 1e-5 && std::fabs(r.cost - opt.result.cost) <= 1e-7 * std::fabs(r.cost));
        for (int i = 0; i <= ph; ++i) for (int j = 0; j < nx; ++j) REQUIRE(std::fabs(seq.state(i, j) - opt.sequence.state(i, j)) <= 2e-5);
        for (int i = 0; i <= ph; ++i) REQUIRE(opt.sequence.input(i, 0) <= 0.5 + 1e-9);
        std::printf("run(): cmd = %.8f (front-end with the built-in system: %.8f), cost %.8f\n", opt.result.cmd(0), r.cmd(0), opt.result.cost);

        // status mapping (NLOptimizer.hpp:729-750): a stop on the iteration count is MAX_ITERATION with a usable command ...
        mpc::NLParameters few = prm;
        few.maximum_iteration = 2;
        opt.setParameters(few);
        opt.run(x0, u0);
        REQUIRE(opt.result.status == mpc::ResultStatus::MAX_ITERATION && opt.result.solver_status == 5);
        opt.setParameters(prm);
        // ... bounds through NLOptimizer::setInputBounds / setStateBounds; an invalid slice is refused, not applied
        mpc::cvec<> ulo(nu, 1), uhi(nu, 1), xlo(nx, 1), xhi(nx, 1);
        ulo(0) = -0.3; uhi(0) = 0.3;
        REQUIRE(opt.setInputBounds(ulo, uhi, mpc::HorizonSlice::all()));
        REQUIRE(!opt.setInputBounds(ulo, uhi, mpc::HorizonSlice{4, 2}));
        opt.run(x0, u0);
        REQUIRE(opt.result.status == mpc::ResultStatus::SUCCESS);
        for (int i = 0; i < ph; ++i) REQUIRE(std::fabs(opt.sequence.input(i, 0)) <= 0.3 + 1e-9);
        const double cmd_bounded = opt.result.cmd(0);
        // ... and a state box the start cannot stay inside ends as ERROR with the previous command and an infinite cost (:611-624)
        xlo(0) = -0.05; xhi(0) = 0.05; xlo(1) = -0.05; xhi(1) = 0.05;
        REQUIRE(opt.setStateBounds(xlo, xhi, mpc::HorizonSlice::all()));
        opt.run(x0, u0);
        REQUIRE(opt.result.status == mpc::ResultStatus::ERROR && opt.result.cmd(0) == u0(0) && std::isinf(opt.result.cost));
        xlo(0) = xlo(1) = -mpc::inf; xhi(0) = xhi(1) = mpc::inf;
        REQUIRE(opt.setStateBounds(xlo, xhi, mpc::HorizonSlice::all()));
        opt.run(x0, u0);
        REQUIRE(opt.result.status == mpc::ResultStatus::SUCCESS && std::fabs(opt.result.cmd(0) - cmd_bounded) <= 1e-9);

        // runBatch: 128 perturbed initial states, device pointers
        const int Bn = 128;
        std::vector<double> hx((size_t)Bn * nx), hu((size_t)Bn * nu, 0.0), hc((size_t)Bn * nu);
        for (int b = 0; b < Bn; ++b) { hx[2 * b] = x0(0) + (b ? 0.4 * std::sin(0.37 * b) : 0.0); hx[2 * b + 1] = x0(1) + (b ? 0.3 * std::cos(0.21 * b) : 0.0); }
        double *dx = nullptr, *du = nullptr, *dc = nullptr; int32_t *ds = nullptr;
        REQUIRE(hipMalloc((void **)&dx, 8 * hx.size()) == hipSuccess && hipMalloc((void **)&du, 8 * hu.size()) == hipSuccess);
        REQUIRE(hipMalloc((void **)&dc, 8 * hc.size()) == hipSuccess && hipMalloc((void **)&ds, 4 * Bn) == hipSuccess);
        REQUIRE(hipMemcpy(dx, hx.data(), 8 * hx.size(), hipMemcpyHostToDevice) == hipSuccess);
        REQUIRE(hipMemcpy(du, hu.data(), 8 * hu.size(), hipMemcpyHostToDevice) == hipSuccess);
        mpcx_nlmpc_batch bt{};
        bt.batch = Bn; bt.x0 = dx; bt.u0 = du; bt.cmd = dc; bt.status = ds;
        REQUIRE(opt.runBatch(bt, nullptr) == MPCX_OK);
        REQUIRE(hipDeviceSynchronize() == hipSuccess);
        std::vector<int32_t> hs(Bn);
        REQUIRE(hipMemcpy(hc.data(), dc, 8 * hc.size(), hipMemcpyDeviceToHost) == hipSuccess);
        REQUIRE(hipMemcpy(hs.data(), ds, 4 * Bn, hipMemcpyDeviceToHost) == hipSuccess);
        int solved = 0;
        for (int b = 0; b < Bn; ++b) solved += hs[b] == 0;
        REQUIRE(solved >= Bn - 2);
        REQUIRE(hc[0] == cmd_bounded);                                           // row 0 is the instance run() solved
        (void)hipFree(dx); (void)hipFree(du); (void)hipFree(dc); (void)hipFree(ds);
        std::printf("runBatch(): %d of %d solved\n", solved, Bn);
    }
    std::printf(failures ? "%d failure(s)\n" : "all NLOptimizer backend checks passed\n", failures);
    return failures ? 1 : 0;
}
