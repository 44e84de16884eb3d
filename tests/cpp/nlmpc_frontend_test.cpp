// The reference's Van der Pol example (examples/vanderpol_ex.cpp) written against this repository's mpc::NLMPC<>:
// same template arguments, same parameter struct, same closed loop; the three closure setters are replaced by
// setModel(MPCX_MODEL_VANDERPOL) (see include/mpcx/NLMPC.hpp for why).
#include <cmath>
#include <cstdio>
#include <cstring>

#include <mpc/NLMPC.hpp>

#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

constexpr int num_states = 2, num_output = 2, num_inputs = 1, pred_hor = 10, ctrl_hor = 5, ineq_c = pred_hor + 1, eq_c = 0;

static int api()
{
    mpc::NLMPC<num_states, num_inputs, num_output, pred_hor, ctrl_hor, ineq_c, eq_c> c;
    CHECK(c.setLoggerLevel(mpc::Logger::LogLevel::NORMAL));
    CHECK(c.setDiscretizationSamplingTime(0.1));
    bool threw = false;
    try { c.setObjectiveFunction([](const auto &, const auto &, const auto &, double) { return 0.0; }); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
    threw = false;
    try { mpc::cvec<2> a, b; c.setOutputBounds(a, b); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
    threw = false;
    try { mpc::cvec<2> x; mpc::cvec<1> u; x.setZero(); u.setZero(); c.optimize(x, u); } catch (const std::runtime_error &) { threw = true; }   // no model yet
    CHECK(threw);
    mpc::NLParameters p;
    CHECK(p.maximum_iteration == 100 && p.relative_ftol == -1 && p.hard_constraints && !p.enable_warm_start);      // Types.hpp:108-143
    std::printf("all C++ NLMPC front-end checks passed (api)\n");
    return 0;
}

static int solve()
{
    mpc::NLMPC<num_states, num_inputs, num_output, pred_hor, ctrl_hor, ineq_c, eq_c> controller;
    controller.setLoggerLevel(mpc::Logger::LogLevel::NORMAL);
    const double ts = 0.1;
    controller.setDiscretizationSamplingTime(ts);
    mpc::NLParameters params;
    params.maximum_iteration = 1000;
    controller.setOptimizerParameters(params);
    controller.setModel(MPCX_MODEL_VANDERPOL);

    mpc::cvec<num_states> modelX, modeldX;
    modelX(0) = 0; modelX(1) = 1.0;
    auto r = controller.getLastResult();
    r.cmd.setZero();
    int steps = 0;
    double first_cmd = 0;
    for (;;) {                                                        // vanderpol_ex.cpp:76-85
        r = controller.optimize(modelX, r.cmd);
        CHECK(r.status == mpc::ResultStatus::SUCCESS && r.is_feasible);
        if (steps == 0) first_cmd = r.cmd(0);
        modeldX(0) = ((1.0 - (modelX(1) * modelX(1))) * modelX(0)) - modelX(1) + r.cmd(0);
        modeldX(1) = modelX(0);
        modelX(0) += modeldX(0) * ts; modelX(1) += modeldX(1) * ts;
        ++steps;
        if (std::fabs(modelX(0)) <= 1e-2 && std::fabs(modelX(1)) <= 1e-1) break;
        CHECK(steps < 400);
    }
    std::printf("closed loop converged in %d steps, first cmd %.9f\n", steps, first_cmd);
    CHECK(std::fabs(first_cmd - 0.09098444) < 2e-6);                  // the SLSQP oracle's first move (tests/test_nlmpc_gpu.py)
    auto seq = controller.getOptimalSequence();
    CHECK(seq.state.rows() == pred_hor + 1 && seq.input.rows() == pred_hor + 1);
    for (int i = 0; i <= pred_hor; ++i) CHECK(seq.input(i, 0) <= 0.5 + 1e-9);

    // bounds through the front-end, dynamic sizes, warm start, and a batch
    mpc::NLMPC<> dyn(2, 1, 2, 10, 5, 11, 0);
    dyn.setDiscretizationSamplingTime(ts);
    mpc::NLParameters pw; pw.maximum_iteration = 200; pw.enable_warm_start = true;
    dyn.setOptimizerParameters(pw);
    mpc::cvec<mpc::Dynamic> lo(1), hi(1);
    lo(0) = -0.05; hi(0) = 0.05;
    CHECK(dyn.setInputBounds(lo, hi, mpc::HorizonSlice::all()));
    dyn.setModel(MPCX_MODEL_VANDERPOL);
    CHECK(!dyn.setInputBounds(lo, hi, mpc::HorizonSlice(2, 9)));     // beyond the control horizon
    mpc::cvec<mpc::Dynamic> x0(2), u0(1);
    x0(0) = 0; x0(1) = 1; u0(0) = 0;
    auto rb = dyn.optimize(x0, u0);
    CHECK(rb.status == mpc::ResultStatus::SUCCESS && std::fabs(rb.cmd(0)) <= 0.05 + 1e-9);
    auto rb2 = dyn.optimize(x0, rb.cmd);                              // warm: the shifted previous solution
    CHECK(rb2.status == mpc::ResultStatus::SUCCESS && std::fabs(rb2.cmd(0) - rb.cmd(0)) < 1e-5);
    const double X0[6] = {0, 1, 0.2, -0.3, -0.5, 0.4}, U0[3] = {0, 0, 0};
    auto R = dyn.optimizeBatch(3, X0, U0);
    CHECK(R.status[0] == 0 && R.status[1] == 0 && R.status[2] == 0);
    CHECK(std::fabs(R.cmd[0] - rb.cmd(0)) < 1e-6);
    std::printf("all C++ NLMPC front-end checks passed (solve)\n");
    return 0;
}

int main(int argc, char **argv)
{
    try {
        if (argc > 1 && !std::strcmp(argv[1], "api")) return api();
        return solve();
    } catch (const std::exception &e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
}
