// TEST INFRASTRUCTURE -- runs the NLMPC kernels of include/mpcx/ through the lock-step interpreter of tests/emu/hip/hip_runtime.h
// on the host (no GPU, nothing of libmpcx.so): one process = one controller set-up and a batch of instances read from stdin.
//
//   run_nlmpc <model> <ph> <ch> <Ts> <hard> <max_iter> <form> [key=value ...] < instances
//     model: vanderpol | vanderpol_terminal | ugv | osc6 | osc8;  form: wave (nlmpc_sqp, one wavefront per instance) | wg (workgroup form)
//     keys: lbu= ubu= (scalar input bounds on every block), lbx0= ubx0= (bounds on state component 0, every step; xs=<first state row
//           that carries them>; HIPEMU_BLOCKS=1|0 in the environment: folded blocks and reduced rows of the workgroup form in LDS | workspace),
//           warm=1 (second solve from the shifted solution), su=, ss= (uniform scalings)
//   stdin: one instance per line: x0[nx] u0[nu]
//   stdout: one JSON object per instance
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "mpcx/nlmpc_engine.hpp"
#ifdef HIPEMU_WITH_WG
#ifdef HIPEMU_CHECK_CARRY
#include "wg_probes.hpp"
#endif
#include "mpcx/nlmpc_sqp_wg.hpp"
#endif

namespace mpcx { namespace engine {
alignas(64) double smem[40960];
extern double lds_base[] __attribute__((alias("_ZN4mpcx6engine4smemE")));
} }

using namespace mpcx;

template <class Mdl>
static int run(int argc, char **argv)
{
    const int ph = atoi(argv[2]), ch = atoi(argv[3]);
    const double Ts = atof(argv[4]);
    const int hard = atoi(argv[5]), max_iter = atoi(argv[6]);
    const std::string form = argv[7];
    double lbu = -INFINITY, ubu = INFINITY, lbx0 = -INFINITY, ubx0 = INFINITY, suv = 1.0, ssv = 1.0;
    int xs = 0;
    int warm = 0;
    std::vector<double> prm_override;
    for (int a = 8; a < argc; ++a) {
        std::string kv = argv[a];
        const size_t eq = kv.find('=');
        const std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
        if (k == "lbu") lbu = atof(v.c_str()); else if (k == "ubu") ubu = atof(v.c_str());
        else if (k == "lbx0") lbx0 = atof(v.c_str()); else if (k == "ubx0") ubx0 = atof(v.c_str());
        else if (k == "xs") xs = atoi(v.c_str()); else if (k == "warm") warm = atoi(v.c_str()); else if (k == "su") suv = atof(v.c_str()); else if (k == "ss") ssv = atof(v.c_str());
        else { fprintf(stderr, "unknown key %s\n", k.c_str()); return 2; }
    }
    constexpr int NX = Mdl::NX, NU = Mdl::NU;
    NlmpcDev M{};
    M.nx = NX; M.nu = NU; M.ny = Mdl::NY; M.ph = ph; M.ch = ch; M.nineq = Mdl::nineq(ph); M.nue = Mdl::neq_user(ph); M.Ts = Ts;
    std::vector<double> prm;
    if (std::string(argv[1]) == "ugv") prm = {0.7071067811865476, 0.7071067811865476, 2.0, 1.0, 0.3, 1.0, 1.0, 0.3, Ts};
    else if (std::string(argv[1]).substr(0, 3) == "osc") prm = {1.0, 0.1};
    else if (std::string(argv[1]) == "vanderpol_rate") prm = {0.1};
    else prm = {0.0};
    M.params = prm.data();
    std::vector<double> su(NU, suv), ss(NX, ssv), iss(NX, 1.0 / ssv);
    M.su = su.data(); M.ss = ss.data(); M.iss = iss.data(); M.scaled = (suv != 1.0 || ssv != 1.0) ? 1 : 0;
    M.nbnd = 0;
    engine::nlmpc_plan(M);
    const int nz = M.nz, nxs = ph * NX;
    std::vector<double> lb(nz, -INFINITY), ub(nz, INFINITY);
    for (int i = xs; i < ph; ++i) { lb[i * NX] = lbx0; ub[i * NX] = ubx0; }      // (xs: the first state row that carries the bound)
    for (int k = 0; k < ch * NU; ++k) { lb[nxs + k] = lbu; ub[nxs + k] = ubu; }
    std::vector<int> bidx; std::vector<double> bsign, bval;
    for (int k = 0; k < nz - 1; ++k) {
        if (ub[k] < 1e30) { bidx.push_back(k); bsign.push_back(1.0); bval.push_back(ub[k]); }
        if (lb[k] > -1e30) { bidx.push_back(k); bsign.push_back(-1.0); bval.push_back(lb[k]); }
    }
    bidx.push_back(0); bsign.push_back(0); bval.push_back(0);
    M.zlb = lb.data(); M.zub = ub.data(); M.nbnd = (int)bidx.size() - 1; M.nbnd_state = 0; for (int k = 0; k < M.nbnd; ++k) M.nbnd_state += bidx[k] < nxs ? 1 : 0; M.bnd_idx = bidx.data(); M.bnd_sign = bsign.data(); M.bnd_val = bval.data();
    engine::nlmpc_plan(M);

    std::vector<double> X0, U0;
    for (;;) {
        std::vector<double> row(NX + NU);
        bool ok = true;
        for (double &v : row) ok = ok && scanf("%lf", &v) == 1;
        if (!ok) break;
        X0.insert(X0.end(), row.begin(), row.begin() + NX); U0.insert(U0.end(), row.begin() + NX, row.end());
    }
    const int B = (int)(X0.size() / NX);
    const int mt = M.nineq + M.nue + M.nbnd;
    size_t ws_total = M.ws.total;
#ifdef HIPEMU_WITH_WG
    engine::WgPlan P{};
    if (form == "wg") {
        int nsb = 0;
        for (int k = 0; k < M.nbnd; ++k) nsb += bidx[k] < nxs ? 1 : 0;
        const int waves = getenv("HIPEMU_WAVES") ? atoi(getenv("HIPEMU_WAVES")) : 0;
        auto envi = [](const char *k) { const char *e = getenv(k); return e ? atoi(e) : -1; };      // (as the launcher reads them when a handle is created)
        if (engine::wg_plan<Mdl>(M, hard, waves, nsb, P, envi("HIPEMU_BLOCKS"), true, 160 * 1024, envi("MPCX_NLMPC_MINV"), envi("MPCX_NLMPC_CARRY"), envi("MPCX_NLMPC_CURV0"), envi("MPCX_NLMPC_CURV0_IT"), envi("MPCX_NLMPC_INV_NB")) != 0) { fprintf(stderr, "the workgroup form does not take this shape\n"); return 3; }
        if (getenv("HIPEMU_VERBOSE")) fprintf(stderr, "wg plan: waves %d, lds %d doubles (%.1f KB), kw %d, nd %d, nsx %d, ws %zu doubles, inverse form %d carried %d (plan %d of %d doubles)\n", P.waves, P.lds_total, P.lds_total / 128.0, P.kw, P.nd, P.nsx, ws_total, P.minv, P.carry_m, P.ws_total, M.ws.scal);
    }
#endif
    // (workspace and LDS start as NaNs: a kernel that reads what it has not written shows it -- the interpreter's memory would otherwise be zeros)
    std::fill(engine::smem, engine::smem + 40960, std::nan(""));
    std::vector<double> ws((size_t)B * ws_total, std::nan("")), cmd(B * NU), cost(B), zout((size_t)B * nz), sx((size_t)B * (ph + 1) * NX), su_((size_t)B * (ph + 1) * NU),
        sy((size_t)B * (ph + 1) * (Mdl::NY > 0 ? Mdl::NY : 1)), mu((size_t)B * (mt > 0 ? mt : 1));
    std::vector<int> status(B), sstat(B), feas(B), iters(B);
    NlmpcSolveDev S{};
    S.batch = B; S.x0 = X0.data(); S.u0 = U0.data(); S.z_warm = nullptr; S.ws = ws.data(); S.max_iter = max_iter; S.hard = hard;
    S.keep_curvature = 0; S.tol_step = 1e-6; S.tol_con = 1e-8; S.ieq_tol = 1e-10; S.eq_tol = 1e-10;
    S.ftol_rel = S.ftol_abs = S.xtol_rel = S.xtol_abs = -1.0;
    S.cmd = cmd.data(); S.cost = cost.data(); S.z_out = zout.data(); S.status = status.data(); S.solver_status = sstat.data();
    S.is_feasible = feas.data(); S.iterations = iters.data(); S.seq_state = sx.data(); S.seq_input = su_.data(); S.seq_output = sy.data();
    S.mu_out = mu.data();
    // form "wave-blk2": the one-wavefront kernel with the dynamics blocks in LDS AND a two-level factor -- a combination the product's plan
    // never selects (it returned wrong results on the GPU for a reason that was not found); here to look for that reason
    const bool blk2 = form == "wave-blk2";
    if (blk2) {
        const int jl = (ph * NX * (2 * NX + NU) + ph * NX * NX + 2 * ph * NX + M.nr + 1) & ~1;
        if (M.lds_blocks < 0) { M.lds_blocks = M.lds_per_wave; M.lds_per_wave += jl; }
        if (getenv("HIPEMU_VERBOSE")) fprintf(stderr, "wave-blk2: kw %d nl %d lds_per_wave %d blocks at %d\n", M.kw, M.nl, M.lds_per_wave, M.lds_blocks);
    }
    auto solve = [&]() {
        int rc;
        if (blk2) {
            if constexpr (engine::kSqpLdsBlocks<Mdl>) {
                const int wpb = nlmpc_waves_per_block(M);
                hipLaunchKernelGGL((engine::nlmpc_sqp<Mdl, true, true>), dim3((S.batch + wpb - 1) / wpb), dim3(wpb * 64), 0, nullptr, M, S);
                return;
            } else { fprintf(stderr, "no LDS-block form for this model\n"); exit(5); }
        }
#ifdef HIPEMU_WITH_WG
        if (form == "wg") rc = engine::launch_solve_wg<Mdl>(&M, &S, &P, nullptr);
        else
#endif
            rc = engine::launch_solve<Mdl>(nullptr, &M, &S, nullptr);
        if (rc != 0) { fprintf(stderr, "launch failed: %d\n", rc); exit(4); }
    };
    solve();
    std::vector<double> zprev;
    for (int pass = 0; pass <= warm; ++pass) {
        if (pass == 1) { zprev = zout; S.z_warm = zprev.data(); S.keep_curvature = warm > 1 ? 1 : 0; solve(); }
        for (int b = 0; b < B; ++b) {
            printf("{\"pass\": %d, \"b\": %d, \"status\": %d, \"solver_status\": %d, \"feasible\": %d, \"iterations\": %d, \"cost\": %.17g, \"cmd\": [", pass, b,
                   status[b], sstat[b], feas[b], iters[b], std::isfinite(cost[b]) ? cost[b] : 1e308);
            for (int j = 0; j < NU; ++j) printf("%s%.17g", j ? ", " : "", cmd[b * NU + j]);
            printf("], \"z\": [");
            for (int k = 0; k < nz; ++k) printf("%s%.17g", k ? ", " : "", zout[(size_t)b * nz + k]);
            auto stat = [&](int k) { const double x = ws[(size_t)b * ws_total + M.ws.scal + k]; return std::isfinite(x) ? x : -1.0; };       // (the one-wavefront form files fewer)
            printf("], \"max_nw\": %g, \"dual_steps\": %g, \"shed\": %g, \"carried\": %g, \"attempts\": %d, \"left_inverse\": %d, \"mu\": [", stat(12), stat(1), stat(13), stat(15), form == "wg" ? 1 + (((int)stat(14) >> 1) & 1) : 1, form == "wg" ? ((int)stat(14) >> 2) & 1 : 0);
            for (int k = 0; k < mt; ++k) printf("%s%.17g", k ? ", " : "", mu[(size_t)b * mt + k]);
            printf("]}\n");
        }
    }
    if (getenv("HIPEMU_VERBOSE")) fprintf(stderr, "block syncs %ld, wave syncs %ld\n", hipemu::st().n_block_syncs, hipemu::st().n_wave_syncs);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 8) { fprintf(stderr, "usage: run_nlmpc model ph ch Ts hard max_iter form [key=value ...]\n"); return 2; }
    const std::string m = argv[1];
    using namespace mpcx::models;
    if (m == "vanderpol") return run<VanDerPol>(argc, argv);
    if (m == "vanderpol_terminal") return run<VanDerPolTerminal>(argc, argv);
    if (m == "vanderpol_rate") return run<VanDerPolRate>(argc, argv);
    if (m == "ugv") return run<Ugv>(argc, argv);
    if (m == "osc6") return run<Oscillators<6>>(argc, argv);
    if (m == "osc8") return run<Oscillators<8>>(argc, argv);
    fprintf(stderr, "unknown model %s\n", m.c_str());
    return 2;
}
