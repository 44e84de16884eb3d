// The binding of INTEGRATION.md section 2, compiled: a backend of the SHAPE of the reference's optimizer interface
// (include/mpc/IOptimizer.hpp:24-58 -- onInit(), setParameters(const Parameters &), run(x0, u0), the members `result` and
// `sequence`) whose run() is the C ABI of include/mpcx.h.  The reference's own IOptimizer.hpp cannot be included here (it pulls in
// Eigen, which this image does not have), so the interface is declared below with this repository's matrix types; the backend class is
// the one a libmpc++ maintainer would write against the real header, member for member.
//
// Checked: (api, no GPU) a host-only handle accepts every setter the backend forwards and run() reports ERROR instead of throwing;
// (solve, GPU) run() returns the result and the sequences that this repository's mpc::LMPC<> front-end returns for the same
// controller, and runBatch() -- the reason to switch -- solves 256 perturbed instances in one launch with row 0 equal to run().
#include <mpc/LMPC.hpp>
#include <mpcx.h>

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

static int failures = 0;
#define REQUIRE(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

namespace mpc {
// the shape of IOptimizer<sizer> (IOptimizer.hpp:24-58), run-time sizes
class IOptimizerShape {
public:
    virtual ~IOptimizerShape() {}
    void initialize(int nx, int nu, int ndu, int ny, int ph, int ch) { nx_ = nx; nu_ = nu; ndu_ = ndu; ny_ = ny; ph_ = ph; ch_ = ch; onInit(); }
    virtual void onInit() = 0;
    virtual void setParameters(const Parameters &param) = 0;
    virtual void run(const cvec<> &x0, const cvec<> &u0) = 0;
    Result<> result;
    OptSequence<> sequence;

protected:
    int nx() const { return nx_; }
    int nu() const { return nu_; }
    int ndu() const { return ndu_; }
    int ny() const { return ny_; }
    int ph() const { return ph_; }
    int ch() const { return ch_; }

private:
    int nx_ = 0, nu_ = 0, ndu_ = 0, ny_ = 0, ph_ = 0, ch_ = 0;
};

class MpcxOptimizer : public IOptimizerShape {
    mpcx_lmpc_t h_ = nullptr;
    int device_;

public:
    explicit MpcxOptimizer(int device) : device_(device) {}
    ~MpcxOptimizer() override { if (h_) mpcx_lmpc_destroy(h_); }
    void onInit() override
    {
        mpcx_dims d{nx(), nu(), ndu(), ny(), ph(), ch()};
        if (mpcx_lmpc_create(&d, device_, &h_) != MPCX_OK) throw std::runtime_error(mpcx_last_error());
        result.cmd.resize(nu(), 1);
        sequence.state.resize(ph() + 1, nx()); sequence.output.resize(ph() + 1, ny()); sequence.input.resize(ph() + 1, nu());
    }
    void setParameters(const Parameters &p) override
    {
        const auto &lp = dynamic_cast<const LParameters &>(p);
        mpcx_lparams q{lp.maximum_iteration, lp.time_limit, lp.enable_warm_start, lp.alpha, lp.rho, lp.eps_rel, lp.eps_abs,
                       lp.eps_prim_inf, lp.eps_dual_inf, lp.verbose, lp.adaptive_rho, lp.polish};
        if (mpcx_lmpc_set_optimizer_parameters(h_, &q) != MPCX_OK) throw std::runtime_error(mpcx_last_error());
    }
    // what LMPC.hpp's setters forward to (ProblemBuilder's in the reference): one line per row of INTEGRATION.md section 1
    bool setStateModel(const mat<> &A, const mat<> &B, const mat<> &C) { return mpcx_lmpc_set_state_space_model(h_, A.data(), B.data(), C.data()) == MPCX_OK; }
    bool setObjective(const cvec<> &ow, const cvec<> &uw, const cvec<> &duw, const HorizonSlice &s)
    {
        return mpcx_lmpc_set_objective_weights_slice(h_, ow.data(), uw.data(), duw.data(), s.start, s.end) == MPCX_OK;
    }
    bool setInputBounds(const cvec<> &lo, const cvec<> &hi, const HorizonSlice &s) { return mpcx_lmpc_set_input_bounds_slice(h_, lo.data(), hi.data(), s.start, s.end) == MPCX_OK; }
    bool setStateBounds(const cvec<> &lo, const cvec<> &hi, const HorizonSlice &s) { return mpcx_lmpc_set_state_bounds_slice(h_, lo.data(), hi.data(), s.start, s.end) == MPCX_OK; }
    bool setReferences(const cvec<> &y, const cvec<> &u, const cvec<> &du, const HorizonSlice &s)
    {
        return mpcx_lmpc_set_references_slice(h_, y.data(), u.data(), du.data(), s.start, s.end) == MPCX_OK;
    }

    void run(const cvec<> &x0, const cvec<> &u0) override          // IOptimizer.hpp:50
    {
        const int n1 = ph() + 1;
        std::vector<double> ss((size_t)n1 * nx()), so((size_t)n1 * ny()), si((size_t)n1 * nu());
        int32_t st = 4, sst = 0, feas = 0;
        if (mpcx_lmpc_solve_host(h_, 1, x0.data(), u0.data(), result.cmd.data(), &result.cost, &st, &sst, &feas, ss.data(), so.data(),
                                 si.data()) != MPCX_OK) {
            result.status = ResultStatus::ERROR;                       // as LOptimizer does when osqp_setup / osqp_solve fail
            result.solver_status_msg = mpcx_last_error();
            return;
        }
        result.status = static_cast<ResultStatus>(st); result.solver_status = sst; result.is_feasible = feas != 0;
        for (int i = 0; i < n1; ++i) {                                  // row-major [(ph+1) x n] from the device
            for (int j = 0; j < nx(); ++j) sequence.state(i, j) = ss[(size_t)i * nx() + j];
            for (int j = 0; j < ny(); ++j) sequence.output(i, j) = so[(size_t)i * ny() + j];
            for (int j = 0; j < nu(); ++j) sequence.input(i, j) = si[(size_t)i * nu() + j];
        }
    }
    // the reason to switch: B controllers' worth of run() in one launch, device pointers in, device pointers out
    int runBatch(const mpcx_lmpc_batch &b, hipStream_t s) { return mpcx_lmpc_solve_batch(h_, &b, s); }
};
}  // namespace mpc

// a chain of two double integrators (positions, velocities), sampled at 0.1 s: nothing of the reference's examples
static void model(mpc::mat<> &A, mpc::mat<> &B, mpc::mat<> &C)
{
    A.resize(4, 4); B.resize(4, 2); C.resize(2, 4);
    A.setIdentity(); A(0, 2) = 0.1; A(1, 3) = 0.1; A(1, 0) = 0.02;
    B.setZero(); B(0, 0) = 0.005; B(2, 0) = 0.1; B(1, 1) = 0.005; B(3, 1) = 0.1;
    C.setZero(); C(0, 0) = 1.0; C(1, 1) = 1.0;
}

int main(int argc, char **argv)
{
    const bool solve = argc > 1 && std::strcmp(argv[1], "solve") == 0;
    const int nx = 4, nu = 2, ny = 2, ph = 12, ch = 6;
    mpc::mat<> A, B, C;
    model(A, B, C);
    mpc::cvec<> ow(ny, 1), uw(nu, 1), duw(nu, 1), umin(nu, 1), umax(nu, 1), xmin(nx, 1), xmax(nx, 1), yref(ny, 1), uref(nu, 1), duref(nu, 1);
    ow.setOnes(); ow *= 4.0; uw.setOnes(); uw *= 0.05; duw.setOnes(); duw *= 0.2;
    umin.setOnes(); umin *= -0.6; umax.setOnes(); umax *= 0.6;
    xmin.setOnes(); xmin *= -mpc::inf; xmax.setOnes(); xmax *= mpc::inf; xmax(2) = 0.35; xmin(2) = -0.35;
    yref(0) = 1.0; yref(1) = -0.5; uref.setZero(); duref.setZero();
    mpc::LParameters prm;
    prm.maximum_iteration = 250;

    mpc::MpcxOptimizer opt(solve ? 0 : -1);
    opt.initialize(nx, nu, 0, ny, ph, ch);
    opt.setParameters(prm);
    REQUIRE(opt.setStateModel(A, B, C));
    REQUIRE(opt.setObjective(ow, uw, duw, mpc::HorizonSlice::all()));
    REQUIRE(opt.setInputBounds(umin, umax, mpc::HorizonSlice::all()));
    REQUIRE(opt.setStateBounds(xmin, xmax, mpc::HorizonSlice::all()));
    REQUIRE(opt.setReferences(yref, uref, duref, mpc::HorizonSlice::all()));
    REQUIRE(!opt.setInputBounds(umin, umax, mpc::HorizonSlice{3, 2}));            // an invalid slice is refused, not applied
    mpc::cvec<> x0(nx, 1), u0(nu, 1);
    x0.setZero(); x0(0) = 0.2; x0(3) = -0.1; u0.setZero();

    if (!solve) {
        opt.run(x0, u0);                                                          // no device behind this handle
        REQUIRE(opt.result.status == mpc::ResultStatus::ERROR);
        REQUIRE(!opt.result.solver_status_msg.empty());
    } else {
        opt.run(x0, u0);
        REQUIRE(opt.result.status == mpc::ResultStatus::SUCCESS);
        // the same controller through this repository's mpc::LMPC<> front-end
        mpc::LMPC<> ref(nx, nu, 0, ny, ph, ch);
        ref.setOptimizerParameters(prm);
        REQUIRE(ref.setStateSpaceModel(A, B, C));
        REQUIRE(ref.setObjectiveWeights(ow, uw, duw, mpc::HorizonSlice::all()));
        REQUIRE(ref.setInputBounds(umin, umax, mpc::HorizonSlice::all()));
        REQUIRE(ref.setStateBounds(xmin, xmax, mpc::HorizonSlice::all()));
        REQUIRE(ref.setReferences(yref, uref, duref, mpc::HorizonSlice::all()));
        const auto r = ref.optimize(x0, u0);
        const auto seq = ref.getOptimalSequence();
        REQUIRE(r.status == opt.result.status && r.solver_status == opt.result.solver_status && r.is_feasible == opt.result.is_feasible);
        REQUIRE(r.cost == opt.result.cost);
        for (int j = 0; j < nu; ++j) REQUIRE(r.cmd(j) == opt.result.cmd(j));
        for (int i = 0; i <= ph; ++i) {
            for (int j = 0; j < nx; ++j) REQUIRE(seq.state(i, j) == opt.sequence.state(i, j));
            for (int j = 0; j < nu; ++j) REQUIRE(seq.input(i, j) == opt.sequence.input(i, j));
        }
        for (int i = 1; i <= ph; ++i) {                                            // the plan respects what was asked for
            for (int j = 0; j < nu; ++j) REQUIRE(std::fabs(opt.sequence.input(i, j)) <= 0.6 + 1e-9);
            REQUIRE(std::fabs(opt.sequence.state(i, 2)) <= 0.35 + 1e-7);
        }
        std::printf("run(): cmd = [%.6f %.6f], cost %.6f\n", opt.result.cmd(0), opt.result.cmd(1), opt.result.cost);

        // runBatch: 256 perturbed initial states, device pointers
        const int Bn = 256;
        std::vector<double> hx((size_t)Bn * nx), hu((size_t)Bn * nu, 0.0), hc((size_t)Bn * nu);
        for (int b = 0; b < Bn; ++b) for (int j = 0; j < nx; ++j) hx[(size_t)b * nx + j] = x0(j) + (b ? 0.3 * std::sin(0.37 * b + j) : 0.0);
        double *dx = nullptr, *du = nullptr, *dc = nullptr; int32_t *ds = nullptr;
        REQUIRE(hipMalloc((void **)&dx, 8 * hx.size()) == hipSuccess && hipMalloc((void **)&du, 8 * hu.size()) == hipSuccess);
        REQUIRE(hipMalloc((void **)&dc, 8 * hc.size()) == hipSuccess && hipMalloc((void **)&ds, 4 * Bn) == hipSuccess);
        REQUIRE(hipMemcpy(dx, hx.data(), 8 * hx.size(), hipMemcpyHostToDevice) == hipSuccess);
        REQUIRE(hipMemcpy(du, hu.data(), 8 * hu.size(), hipMemcpyHostToDevice) == hipSuccess);
        mpcx_lmpc_batch bt{};
        bt.batch = Bn; bt.x0 = dx; bt.u0 = du; bt.cmd = dc; bt.status = ds;         // references: the ones given to the setters
        REQUIRE(opt.runBatch(bt, nullptr) == MPCX_OK);
        std::vector<int32_t> hs(Bn);
        REQUIRE(hipMemcpy(hc.data(), dc, 8 * hc.size(), hipMemcpyDeviceToHost) == hipSuccess);
        REQUIRE(hipMemcpy(hs.data(), ds, 4 * Bn, hipMemcpyDeviceToHost) == hipSuccess);
        int solved = 0;
        for (int b = 0; b < Bn; ++b) solved += hs[b] == 0;
        REQUIRE(solved == Bn);
        for (int j = 0; j < nu; ++j) REQUIRE(hc[j] == opt.result.cmd(j));           // row 0 is the instance run() solved
        (void)hipFree(dx); (void)hipFree(du); (void)hipFree(dc); (void)hipFree(ds);
        std::printf("runBatch(): %d of %d solved\n", solved, Bn);
    }
    std::printf(failures ? "%d failure(s)\n" : "all IOptimizer backend checks passed\n", failures);
    return failures ? 1 : 0;
}
