// The reference's Van der Pol example (examples/vanderpol_ex.cpp) against this repository's mpc::NLMPC<> with its hooks
// given the reference's way: setStateSpaceFunction / setObjectiveFunction / setIneqConFunction take lambdas with the
// reference's parameter lists and bodies.  Compiled by hipcc (tests/test_cpp_frontend.py); the only edits to the example's
// lambdas are the capture list ([&] -> [=]: a device closure cannot hold references into the host's stack) and, for the
// helper it calls, a functor with a __device__ call operator.  Also exercised: setHooks (all hooks at once, inlined),
// setHookSources (bodies as text, hipRTC), user equalities, an output function, and the built-in model as the yardstick.
// The lambda bodies are the example's on purpose (that a user's existing hooks run unchanged is what is being tested); nothing
// else of the example is here.
#include <cmath>
#include <cstdio>
#include <cstring>

#include <mpc/NLMPC.hpp>

#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

constexpr int num_states = 2;
constexpr int num_output = 2;
constexpr int num_inputs = 1;
constexpr int pred_hor = 10;
constexpr int ctrl_hor = 5;
constexpr int ineq_c = pred_hor + 1;
constexpr int eq_c = 0;

using Ctrl = mpc::NLMPC<num_states, num_inputs, num_output, pred_hor, ctrl_hor, ineq_c, eq_c>;

struct StateEq {      // vanderpol_ex.cpp:33-41 (a lambda there; a functor here so that host and device code can both call it)
    __host__ __device__ void operator()(mpc::cvec<num_states> &dx, const mpc::cvec<num_states> &x, const mpc::cvec<num_inputs> &u) const
    {
        dx(0) = ((1.0 - (x(1) * x(1))) * x(0)) - x(1) + u(0);
        dx(1) = x(0);
    }
};

static void configure(Ctrl &controller, double ts)
{
    controller.setLoggerLevel(mpc::Logger::LogLevel::NONE);
    controller.setDiscretizationSamplingTime(ts);
    mpc::NLParameters params;
    params.maximum_iteration = 1000;
    controller.setOptimizerParameters(params);
}

// the three setter calls of vanderpol_ex.cpp:43-65
static void reference_hooks(Ctrl &controller)
{
    StateEq stateEq;
    controller.setStateSpaceFunction([=] __device__(
                                        mpc::cvec<num_states> &dx,
                                        const mpc::cvec<num_states> &x,
                                        const mpc::cvec<num_inputs> &u,
                                        const unsigned int &)
                                    { stateEq(dx, x, u); });

    controller.setObjectiveFunction([=] __device__(
                                       const mpc::mat<pred_hor + 1, num_states> &x,
                                       const mpc::mat<pred_hor + 1, num_output> &,
                                       const mpc::mat<pred_hor + 1, num_inputs> &u,
                                       double)
                                   { return x.array().square().sum() + u.array().square().sum(); });

    controller.setIneqConFunction([=] __device__(
                                     mpc::cvec<ineq_c> &in_con,
                                     const mpc::mat<pred_hor + 1, num_states> &,
                                     const mpc::mat<pred_hor + 1, num_output> &,
                                     const mpc::mat<pred_hor + 1, num_inputs> &u,
                                     const double &)
                                 {
        for (int i = 0; i < ineq_c; i++) {
            in_con(i) = u(i, 0) - 0.5;
        } });
}

struct Run { double cmd[3], cost[3]; int status[3], solver[3]; };

static int first_moves(Ctrl &c, Run &out)
{
    const double X0[6] = {0, 1, 0.2, -0.3, -0.5, 0.4}, U0[3] = {0, 0, 0};
    auto R = c.optimizeBatch(3, X0, U0);
    for (int b = 0; b < 3; ++b) { out.cmd[b] = R.cmd[b]; out.cost[b] = R.cost[b]; out.status[b] = R.status[b]; out.solver[b] = R.solver_status[b]; }
    return 0;
}

static int closed_loop(Ctrl &controller, double ts, int &steps, double &first_cmd)
{
    StateEq stateEq;
    mpc::cvec<num_states> modelX, modeldX;
    modelX.resize(num_states);
    modeldX.resize(num_states);
    modelX(0) = 0;
    modelX(1) = 1.0;
    auto r = controller.getLastResult();
    steps = 0;
    for (;;) {                                                        // vanderpol_ex.cpp:76-85
        r = controller.optimize(modelX, r.cmd);
        CHECK(r.status == mpc::ResultStatus::SUCCESS && r.is_feasible);
        if (steps == 0) first_cmd = r.cmd(0);
        stateEq(modeldX, modelX, r.cmd);
        modelX += modeldX * ts;
        ++steps;
        if (std::fabs(modelX[0]) <= 1e-2 && std::fabs(modelX[1]) <= 1e-1) break;
        CHECK(steps < 400);
    }
    return 0;
}

static int solve()
{
    const double ts = 0.1;
    // yardstick: the system built into the library
    Ctrl zoo; configure(zoo, ts); zoo.setModel(MPCX_MODEL_VANDERPOL);
    Run rz; CHECK(first_moves(zoo, rz) == 0);
    for (int b = 0; b < 3; ++b) CHECK(rz.status[b] == 0);

    // (1) the reference's setters, one hook at a time
    Ctrl erased; configure(erased, ts); reference_hooks(erased);
    Run re; CHECK(first_moves(erased, re) == 0);
    // (2) all hooks at once
    Ctrl fused; configure(fused, ts);
    {
        StateEq stateEq;
        fused.setHooks([=] __device__(mpc::cvec<num_states> &dx, const mpc::cvec<num_states> &x, const mpc::cvec<num_inputs> &u, const unsigned int &)
                       { stateEq(dx, x, u); },
                       [=] __device__(const mpc::mat<pred_hor + 1, num_states> &x, const mpc::mat<pred_hor + 1, num_output> &,
                                      const mpc::mat<pred_hor + 1, num_inputs> &u, double)
                       { return x.array().square().sum() + u.array().square().sum(); },
                       [=] __device__(mpc::cvec<ineq_c> &in_con, const mpc::mat<pred_hor + 1, num_states> &, const mpc::mat<pred_hor + 1, num_output> &,
                                      const mpc::mat<pred_hor + 1, num_inputs> &u, const double &)
                       { for (int i = 0; i < ineq_c; i++) { in_con(i) = u(i, 0) - 0.5; } });
    }
    Run rf; CHECK(first_moves(fused, rf) == 0);
    // (3) the bodies as text
    Ctrl jit; configure(jit, ts);
    jit.setHookSources("dx(0) = ((1.0 - (x(1) * x(1))) * x(0)) - x(1) + u(0); dx(1) = x(0);",
                       "return x.array().square().sum() + u.array().square().sum();",
                       "for (int i = 0; i < ineq_c; i++) { in_con(i) = u(i, 0) - 0.5; }");
    Run rj; CHECK(first_moves(jit, rj) == 0);

    int identical = 0;
    for (int b = 0; b < 3; ++b) {
        std::printf("instance %d  zoo %.17g  setters %.17g  setHooks %.17g  sources %.17g\n", b, rz.cmd[b], re.cmd[b], rf.cmd[b], rj.cmd[b]);
        CHECK(re.status[b] == 0 && rf.status[b] == 0 && rj.status[b] == 0);
        // the three routes run the same source through the same engine: the same answer up to the last bits (the routes are
        // separate instantiations -- calls through pointers, everything inlined, run-time compiled -- and the compiler contracts a
        // product into a following addition in one and not in the other: a few ulp)
        auto close = [](double a, double c) { return std::fabs(a - c) <= 1e-13 * std::fmax(1.0, std::fabs(a)); };
        CHECK(close(re.cmd[b], rf.cmd[b]) && close(re.cmd[b], rj.cmd[b]) && close(re.cost[b], rf.cost[b]) && close(re.cost[b], rj.cost[b]));
        // the built-in model spells the same functions by hand: the compiler may contract its products differently, and one
        // ulp in the cost is 1e-8 in a forward-difference gradient -- the optimum agrees to that noise, not to the bit
        const double tol = 2e-6 * std::fmax(1.0, std::fabs(rz.cmd[b]));
        CHECK(std::fabs(re.cmd[b] - rz.cmd[b]) <= tol);
        CHECK(std::fabs(re.cost[b] - rz.cost[b]) <= 1e-9 * std::fabs(rz.cost[b]));
        identical += (re.cmd[b] == rz.cmd[b]);
    }
    std::printf("bit-identical to the built-in model: %d of 3 commands\n", identical);

    // the example's closed loop through the reference's setters
    int steps = 0; double first_cmd = 0;
    CHECK(closed_loop(erased, ts, steps, first_cmd) == 0);
    std::printf("closed loop (hooks through the reference's setters) converged in %d steps, first cmd %.9f\n", steps, first_cmd);
    CHECK(std::fabs(first_cmd - 0.09098444) < 2e-6);                  // the SLSQP oracle's first move (tests/test_nlmpc_gpu.py)

    // user equalities and an output function through the setters: terminal constraint x(ph) = 0 (MPCX_MODEL_VANDERPOL_TERMINAL)
    // with the cost written on the outputs y = x
    using CtrlEq = mpc::NLMPC<num_states, num_inputs, num_output, pred_hor, ctrl_hor, ineq_c, 2>;
    CtrlEq zt; zt.setDiscretizationSamplingTime(ts);
    mpc::NLParameters pe; pe.maximum_iteration = 300; zt.setOptimizerParameters(pe);
    zt.setModel(MPCX_MODEL_VANDERPOL_TERMINAL);
    CtrlEq ht; ht.setDiscretizationSamplingTime(ts); ht.setOptimizerParameters(pe);
    {
        StateEq stateEq;
        ht.setStateSpaceFunction([=] __device__(mpc::cvec<num_states> &dx, const mpc::cvec<num_states> &x, const mpc::cvec<num_inputs> &u,
                                                const unsigned int &) { stateEq(dx, x, u); });
        ht.setOutputFunction([=] __device__(mpc::cvec<num_output> &y, const mpc::cvec<num_states> &x, const mpc::cvec<num_inputs> &,
                                            const unsigned int &) { y(0) = x(0); y(1) = x(1); });
        ht.setObjectiveFunction([=] __device__(const mpc::mat<pred_hor + 1, num_states> &, const mpc::mat<pred_hor + 1, num_output> &y,
                                               const mpc::mat<pred_hor + 1, num_inputs> &u, double)
                                { return y.array().square().sum() + u.array().square().sum(); });
        ht.setIneqConFunction([=] __device__(mpc::cvec<ineq_c> &in_con, const mpc::mat<pred_hor + 1, num_states> &,
                                             const mpc::mat<pred_hor + 1, num_output> &, const mpc::mat<pred_hor + 1, num_inputs> &u, const double &)
                              { for (int i = 0; i < ineq_c; i++) in_con(i) = u(i, 0) - 0.5; });
        ht.setEqConFunction([=] __device__(mpc::cvec<2> &eq_con, const mpc::mat<pred_hor + 1, num_states> &x, const mpc::mat<pred_hor + 1, num_inputs> &)
                            { eq_con(0) = x(pred_hor, 0); eq_con(1) = x(pred_hor, 1); });
    }
    mpc::cvec<num_states> x0; x0(0) = 0.1; x0(1) = 0.1;
    mpc::cvec<num_inputs> u0; u0(0) = 0.0;
    auto a = zt.optimize(x0, u0), bq = ht.optimize(x0, u0);
    std::printf("terminal constraint: zoo cmd %.12f status %d (%d), hooks cmd %.12f status %d (%d)\n", a.cmd(0), (int)a.status, a.solver_status,
                bq.cmd(0), (int)bq.status, bq.solver_status);
    CHECK(a.status == mpc::ResultStatus::SUCCESS && bq.status == mpc::ResultStatus::SUCCESS);
    CHECK(std::fabs(a.cmd(0) - bq.cmd(0)) <= 1e-7 * std::fmax(1.0, std::fabs(a.cmd(0))));
    auto sq = ht.getOptimalSequence();
    CHECK(std::fabs(sq.state(pred_hor, 0)) < 1e-6 && std::fabs(sq.state(pred_hor, 1)) < 1e-6);
    std::printf("all C++ NLMPC hook checks passed\n");
    return 0;
}

int main()
{
    try {
        return solve();
    } catch (const std::exception &e) {
        std::printf("exception: %s\n", e.what());
        return 2;
    }
}
