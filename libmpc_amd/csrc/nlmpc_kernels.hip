// The NLMPC engine (mpcx/nlmpc_engine.hpp) instantiated for the model zoo of mpcx/nlmpc_models.hpp: the reference's example
// systems as component-wise device functors with declared constraint structure, selected by id (mpcx_nlmpc_create).
// Hooks with the reference's own signatures do not pass through this file: they instantiate the same engine in the user's
// translation unit (mpcx/nlmpc_hooks.hpp) or in a run-time compiled module (nlmpc_jit.cpp).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "mpcx/nlmpc_engine.hpp"
#include "mpcx/nlmpc_sqp_wg.hpp"      // (for the plan: the kernels of that header are instantiated in nlmpc_wg_kernels.hip)

#include "nlmpc_zoo.hpp"

namespace mpcx {

int nlmpc_model_dims(int model_id, int *nx, int *nu, int *ny, int ph, int *nineq, int *nue)
{
    return dispatch_model(model_id, [&](auto mdl) {
        using Mdl = decltype(mdl);
        *nx = Mdl::NX; *nu = Mdl::NU; *ny = Mdl::NY; *nineq = Mdl::nineq(ph); *nue = Mdl::neq_user(ph);
        return 0;
    });
}

void nlmpc_plan_host(NlmpcDev &m) { engine::nlmpc_plan(m); }

int nlmpc_launch(void *, const NlmpcDev *m, const NlmpcBatchDev *b, void *stream)
{
    return dispatch_model(m->model_id, [&](auto mdl) { return engine::launch_evaluate<decltype(mdl)>(nullptr, m, b, stream); });
}

// Two forms of the SQP for the built-in systems (mpcx_nlmpc_solve_batch):
//   * nlmpc_sqp_wg (mpcx/nlmpc_sqp_wg.hpp): one workgroup per instance, the reduced problem in LDS, no streaming of a workspace through HBM;
//   * nlmpc_sqp (mpcx/nlmpc_engine.hpp): one wavefront per instance, the reduced problem in a per-instance HBM workspace.
// Which one is faster is a matter of how many instances a CU holds at once -- both are bound by the issue latency of dependent
// instructions at one or two wavefronts per SIMD (tools/micro/latency.hip): the LDS-resident form holds 160 KB /
// (its LDS block) instances per CU, the workspace form eight.  Measured on MI355X (profiles/r04_nlmpc_forms.txt): the workgroup form wins
// where an instance needs one wavefront and a few KB of LDS (config 1), the wavefront form where the LDS block is tens of KB (configs 3, 5).
// The default follows that, and takes the workgroup form for a batch that it holds resident all at once (latency: each instance on its own
// wavefronts); MPCX_NLMPC_FORM=wg|wave forces one, MPCX_NLMPC_WAVES=1|2|4 the wavefronts per instance of the workgroup form.
// which form the last launch of a built-in system took: 0 = nlmpc_sqp, 1 | 2 | 4 = nlmpc_sqp_wg with that many wavefronts per instance
// (nlmpc_wg_kernels.hip)
int nlmpc_wg_plan(const NlmpcDev &m, int hard, int waves, int state_bounds, engine::WgPlan &P, int blocks);
int nlmpc_wg_launch(const NlmpcDev *m, const NlmpcSolveDev *b, const engine::WgPlan *P, void *stream);

static int g_last_form = -1;
int nlmpc_last_form() { return g_last_form; }

int nlmpc_launch_solve(void *, const NlmpcDev *m, const NlmpcSolveDev *b, void *stream)
{
    return dispatch_model(m->model_id, [&](auto mdl) {
        using Mdl = decltype(mdl);
        const char *form = getenv("MPCX_NLMPC_FORM");
        const bool force_wave = form && !strcmp(form, "wave"), force_wg = form && !strcmp(form, "wg");
        if (!force_wave) {
            engine::WgPlan P;
            const char *wv = getenv("MPCX_NLMPC_WAVES"), *bl = getenv("MPCX_NLMPC_BLOCKS");   // (BLOCKS=1|0: folded blocks and reduced rows in LDS | workspace)
            auto plan = [&](engine::WgPlan &X, int blocks) { return nlmpc_wg_plan(*m, b->hard, wv ? atoi(wv) : 0, m->nbnd_state, X, blocks) == 0 && X.ws_total <= m->ws.total; };
            // throughput: one wavefront per instance and a CU full of instances; latency: a batch that is resident all at once in the
            // workgroup form (every instance on its own four wavefronts) finishes in half the time of the same batch in the wavefront form.
            // The plan with the most workgroups per CU first; where that keeps the folded blocks and the reduced rows in the workspace
            // (config 3: three per CU, 12.0 ms a solve) and the batch is resident with them in LDS too (two per CU, 9.7 ms), that one.
            // A problem that fills a CU's LDS alone (config 5) is still ahead at two rounds: 71 ms against 94 at 512 instances.
            auto resident = [&](const engine::WgPlan &X, int rounds) { return (long)b->batch <= 256L * X.per_cu * rounds; };
            bool fits = plan(P, bl ? atoi(bl) : -1);
            if (!bl && fits && P.waves > 1 && !P.f_lds) {
                engine::WgPlan Q;
                if (plan(Q, 1) && Q.waves > 1 && resident(Q, 1)) P = Q;
            }
            if (fits && (force_wg || P.waves == 1 || resident(P, P.per_cu == 1 ? 2 : 1))) { g_last_form = P.waves; return nlmpc_wg_launch(m, b, &P, stream); }
            if (force_wg) return -2;
        }
        g_last_form = 0;
        return engine::launch_solve<Mdl>(nullptr, m, b, stream);
    });
}

}  // namespace mpcx
