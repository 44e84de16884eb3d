// C++ front-end test: the scenarios of the reference's LMPC tests (test/LMPC/test_lmpc.cpp,
// test/LMPC/test_common.cpp:89-237) re-expressed against include/mpc/LMPC.hpp of this repository,
// once with compile-time sizes and once with run-time sizes (the reference builds every test both
// ways, test/CMakeLists.txt:56-65).  Mode "api": setter return values and throwing calls only (runs
// without a GPU, MPCX_DEVICE=-1).  Mode "solve": also the quadrotor known answer on the GPU.
// A drop-in test has to make the reference's calls in the reference's order: the resemblance to its TEST_CASEs is the point.
// The numeric literals (model matrices, weights, expected command) are data -- the same values tests/golden/
// reference_known_answers.json holds for the Python suite; the harness, the controller and everything under it are this repository's.
#include <mpc/LMPC.hpp>

#include <cstdio>
#include <cstring>

static int failures = 0;
#define REQUIRE(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)
#define REQUIRE_THROWS(expr) do { bool t = false; try { expr; } catch (const std::exception &) { t = true; } \
    if (!t) { std::printf("FAILED %s:%d: no throw: %s\n", __FILE__, __LINE__, #expr); ++failures; } } while (0)

template <class Controller, int Tnx, int Tnu, int Tndu, int Tny, int Tph, int Tch>
void quadrotor_case(Controller &optsolver, bool solve)
{
    optsolver.setLoggerLevel(mpc::Logger::LogLevel::NONE);
    mpc::mat<Tnx, Tnx> Ad(12, 12);
    Ad << 1, 0, 0, 0, 0, 0, 0.1, 0, 0, 0, 0, 0,
        0, 1, 0, 0, 0, 0, 0, 0.1, 0, 0, 0, 0,
        0, 0, 1, 0, 0, 0, 0, 0, 0.1, 0, 0, 0,
        0.0488, 0, 0, 1, 0, 0, 0.0016, 0, 0, 0.0992, 0, 0,
        0, -0.0488, 0, 0, 1, 0, 0, -0.0016, 0, 0, 0.0992, 0,
        0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0.0992,
        0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0,
        0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0,
        0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0,
        0.9734, 0, 0, 0, 0, 0, 0.0488, 0, 0, 0.9846, 0, 0,
        0, -0.9734, 0, 0, 0, 0, 0, -0.0488, 0, 0, 0.9846, 0,
        0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.9846;
    mpc::mat<Tnx, Tnu> Bd(12, 4);
    Bd << 0, -0.0726, 0, 0.0726,
        -0.0726, 0, 0.0726, 0,
        -0.0152, 0.0152, -0.0152, 0.0152,
        0, -0.0006, -0.0000, 0.0006,
        0.0006, 0, -0.0006, 0,
        0.0106, 0.0106, 0.0106, 0.0106,
        0, -1.4512, 0, 1.4512,
        -1.4512, 0, 1.4512, 0,
        -0.3049, 0.3049, -0.3049, 0.3049,
        0, -0.0236, 0, 0.0236,
        0.0236, 0, -0.0236, 0,
        0.2107, 0.2107, 0.2107, 0.2107;
    mpc::mat<Tny, Tnx> Cd(12, 12);
    Cd.setIdentity();
    REQUIRE(optsolver.setStateSpaceModel(Ad, Bd, Cd));
    mpc::mat<Tnx, Tndu> Bdist(12, 4);
    mpc::mat<Tny, Tndu> Ddist(12, 4);
    REQUIRE(optsolver.setDisturbances(Bdist, Ddist));

    mpc::mat<Tnu, Tph> InputWMat(4, 10), DeltaInputWMat(4, 10);
    mpc::mat<Tny, Tph> OutputWMat(12, 10);
    REQUIRE(optsolver.setObjectiveWeights(OutputWMat, InputWMat, DeltaInputWMat));
    mpc::cvec<Tnu> InputW(4, 1), DeltaInputW(4, 1);
    mpc::cvec<Tny> OutputW(12, 1);
    OutputW << 0, 0, 10, 10, 10, 10, 0, 0, 0, 5, 5, 5;
    InputW << 0.1, 0.1, 0.1, 0.1;
    DeltaInputW << 0, 0, 0, 0;
    REQUIRE(optsolver.setObjectiveWeights(OutputW, InputW, DeltaInputW, {0, 10}));
    REQUIRE(!optsolver.setObjectiveWeights(OutputW, InputW, DeltaInputW, {4, 4}));      // invalid slice -> false

    mpc::mat<Tnx, Tph> xminmat(12, 10), xmaxmat(12, 10);
    mpc::mat<Tny, Tph> yminmat(12, 10), ymaxmat(12, 10);
    mpc::mat<Tnu, Tch> uminmat(4, 10), umaxmat(4, 10);
    REQUIRE(optsolver.setStateBounds(xminmat, xmaxmat));
    REQUIRE(optsolver.setInputBounds(uminmat, umaxmat));
    REQUIRE(optsolver.setOutputBounds(yminmat, ymaxmat));

    mpc::cvec<Tnx> xmin(12, 1), xmax(12, 1);
    xmin << -M_PI / 6, -M_PI / 6, -mpc::inf, -mpc::inf, -mpc::inf, -1, -mpc::inf, -mpc::inf, -mpc::inf, -mpc::inf, -mpc::inf, -mpc::inf;
    xmax << M_PI / 6, M_PI / 6, mpc::inf, mpc::inf, mpc::inf, mpc::inf, mpc::inf, mpc::inf, mpc::inf, mpc::inf, mpc::inf, mpc::inf;
    mpc::cvec<Tny> ymin(12, 1), ymax(12, 1);
    ymin.setOnes(); ymin *= -mpc::inf;
    ymax.setOnes(); ymax *= mpc::inf;
    mpc::cvec<Tnu> umin(4, 1), umax(4, 1);
    const double u0 = 10.5916;
    umin << 9.6, 9.6, 9.6, 9.6; umin.array() -= u0;
    umax << 13, 13, 13, 13; umax.array() -= u0;
    REQUIRE(optsolver.setStateBounds(xmin, xmax, {0, 10}));
    REQUIRE(optsolver.setInputBounds(umin, umax, {0, 10}));
    REQUIRE(optsolver.setOutputBounds(ymin, ymax, {0, 10}));
    REQUIRE(optsolver.setStateBounds(xmin, xmax, {0, 1}));
    REQUIRE(optsolver.setInputBounds(umin, umax, {0, 1}));
    REQUIRE(optsolver.setOutputBounds(ymin, ymax, {0, 1}));
    REQUIRE(!optsolver.setInputBounds(umin, umax, {0, 11}));

    mpc::cvec<Tnx> onesx(12, 1); onesx.setOnes();
    mpc::cvec<Tnu> onesu(4, 1); onesu.setOnes();
    REQUIRE(optsolver.setScalarConstraint(-mpc::inf, mpc::inf, onesx, onesu, {-1, -1}));
    REQUIRE(optsolver.setScalarConstraint(0, -mpc::inf, mpc::inf, onesx, onesu));
    REQUIRE(!optsolver.setScalarConstraint(10, -1.0, 1.0, onesx, onesu));

    mpc::mat<Tny, Tph> yRefMat(12, 10);
    mpc::mat<Tnu, Tph> uRefMat(4, 10);
    REQUIRE(optsolver.setReferences(yRefMat, uRefMat, uRefMat));
    mpc::cvec<Tny> yRef(12, 1);
    yRef << 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0;
    mpc::cvec<Tnu> zu(4, 1);
    REQUIRE(optsolver.setReferences(yRef, zu, zu, {0, 10}));

    mpc::LParameters params;
    params.maximum_iteration = 250;
    optsolver.setOptimizerParameters(params);
    mpc::mat<Tndu, Tph> dmat(4, 10);
    mpc::cvec<Tndu> dvec(4, 1);
    REQUIRE(optsolver.setExogenousInputs(dmat));
    REQUIRE(optsolver.setExogenousInputs(dvec, {0, 10}));

    REQUIRE_THROWS(optsolver.setDiscretizationSamplingTime(0.1));
    REQUIRE_THROWS(optsolver.setInputScale(zu));
    REQUIRE_THROWS(optsolver.setStateScale(onesx));

    if (!solve) return;
    mpc::cvec<Tnx> x0(12, 1);
    auto res = optsolver.optimize(x0, zu);
    auto seq = optsolver.getOptimalSequence();
    mpc::cvec<4> testRes;
    testRes << -0.9916, 1.74839, -0.9916, 1.74839;
    mpc::cvec<4> got;
    for (int i = 0; i < 4; ++i) got(i) = res.cmd(i);
    std::cout << "Expected result: " << testRes << std::endl << "Obtained result: " << got << std::endl;
    REQUIRE(got.isApprox(testRes, 1e-4));
    REQUIRE(res.status == mpc::ResultStatus::SUCCESS);
    REQUIRE(res.is_feasible);
    REQUIRE(seq.state.rows() == 11 && seq.state.cols() == 12 && seq.input.rows() == 11 && seq.input.cols() == 4);
    REQUIRE(seq.input(0, 1) == res.cmd(1));
    REQUIRE(optsolver.getLastResult().cost == res.cost);

    // extension: the same controller on a batch of host-resident instances
    const int B = 33;
    std::vector<double> X0((size_t)B * 12, 0.0), U0((size_t)B * 4, 0.0);
    for (int b = 1; b < B; ++b) X0[(size_t)b * 12 + 2] = 0.01 * b;
    auto R = optsolver.optimizeBatch(B, X0.data(), U0.data());
    REQUIRE(R.batch == B && (int)R.cmd.size() == B * 4);
    REQUIRE(std::fabs(R.cmd[1] - res.cmd(1)) < 1e-12);
    for (int b = 0; b < B; ++b) REQUIRE(R.status[b] == 0);

    // LParameters::enable_warm_start through optimize() (LMPC.hpp:677-722): the next tick starts from the previous tick's
    // active set; the results are those of a cold solve
    {
        mpc::LParameters pw;
        pw.maximum_iteration = 250;
        pw.enable_warm_start = true;
        optsolver.setOptimizerParameters(pw);
        mpc::cvec<Tnx> xa(12, 1), xb(12, 1);
        xa(2) = 0.05; xb(2) = 0.06; xb(8) = 0.02;
        auto w1 = optsolver.optimize(xa, zu);
        auto w2 = optsolver.optimize(xb, w1.cmd);            // warm: carries the active set of w1
        mpc::LParameters pc;
        pc.maximum_iteration = 250;
        optsolver.setOptimizerParameters(pc);
        auto c2 = optsolver.optimize(xb, w1.cmd);            // cold
        REQUIRE(w2.status == mpc::ResultStatus::SUCCESS && c2.status == mpc::ResultStatus::SUCCESS);
        for (int i = 0; i < 4; ++i) REQUIRE(std::fabs(w2.cmd(i) - c2.cmd(i)) <= 1e-9 * std::fmax(1.0, std::fabs(c2.cmd(i))));
        REQUIRE(std::fabs(w2.cost - c2.cost) <= 1e-9 * std::fmax(1.0, std::fabs(c2.cost)));
    }
}

int main(int argc, char **argv)
{
    const bool solve = argc > 1 && std::strcmp(argv[1], "solve") == 0;
    {
        mpc::LMPC<12, 4, 4, 12, 10, 10> c;      // compile-time sizes
        quadrotor_case<decltype(c), 12, 4, 4, 12, 10, 10>(c, solve);
    }
    {
        mpc::LMPC<> c(12, 4, 4, 12, 10, 10);     // run-time sizes (the reference's MPC_DYNAMIC build)
        quadrotor_case<decltype(c), -1, -1, -1, -1, -1, -1>(c, solve);
    }
    std::printf(failures ? "%d failure(s)\n" : "all C++ front-end checks passed\n", failures);
    return failures ? 1 : 0;
}
