// The NLMPC engine (mpcx/nlmpc_engine.hpp) instantiated for the model zoo of mpcx/nlmpc_models.hpp: the reference's example
// systems as component-wise device functors with declared constraint structure, selected by id (mpcx_nlmpc_create).
// Hooks with the reference's own signatures do not pass through this file: they instantiate the same engine in the user's
// translation unit (mpcx/nlmpc_hooks.hpp) or in a run-time compiled module (nlmpc_jit.cpp).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "mpcx/nlmpc_engine.hpp"
#include "mpcx/nlmpc_sqp_wg.hpp"

namespace mpcx {
namespace {

using namespace models;

template <class F>
int dispatch_model(int model_id, F &&fn)
{
    switch (model_id) {
    case 1: return fn(VanDerPol{});
    case 2: return fn(Ugv{});
    case 3: return fn(Oscillators<6>{});
    case 4: return fn(Oscillators<8>{});
    case 5: return fn(VanDerPolTerminal{});
    default: return -1;
    }
}

}  // namespace

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

// The built-in systems solve in the workgroup form (mpcx/nlmpc_sqp_wg.hpp: one workgroup per instance, the reduced problem in LDS).
// A shape its LDS plan does not take (horizons beyond 64 steps, more than 160 KB) goes through nlmpc_sqp, one wavefront per instance;
// MPCX_NLMPC_FORM=wave forces that form (A/B measurements), MPCX_NLMPC_WAVES=1|2|4 the wavefronts per instance.
int nlmpc_launch_solve(void *, const NlmpcDev *m, const NlmpcSolveDev *b, void *stream)
{
    return dispatch_model(m->model_id, [&](auto mdl) {
        using Mdl = decltype(mdl);
        const char *form = getenv("MPCX_NLMPC_FORM");
        if (!(form && !strcmp(form, "wave"))) {
            engine::WgPlan P;
            const char *wv = getenv("MPCX_NLMPC_WAVES");
            if (engine::wg_plan<Mdl>(*m, b->hard, wv ? atoi(wv) : 0, m->nbnd_state, P) == 0 && P.ws_total <= m->ws.total)
                return engine::launch_solve_wg<Mdl>(m, b, &P, stream);
            if (form && !strcmp(form, "wg")) return -2;
        }
        return engine::launch_solve<Mdl>(nullptr, m, b, stream);
    });
}

}  // namespace mpcx
