// nlmpc_sqp_wg (mpcx/nlmpc_sqp_wg.hpp: one workgroup per instance, the reduced problem in LDS) instantiated for the model zoo: its plan and its
// launcher by model id, called by the dispatcher in nlmpc_kernels.hip.  A translation unit of its own: the thirty kernels of this family (five
// systems x 1 / 2 / 4 wavefronts per instance x blocks in LDS / in the workspace) take as long to compile as everything else together.
#include <hip/hip_runtime.h>

#include "mpcx/nlmpc_engine.hpp"
#include "mpcx/nlmpc_sqp_wg.hpp"
#include "nlmpc_zoo.hpp"

namespace mpcx {

int nlmpc_wg_plan(const NlmpcDev &m, int hard, int waves, int state_bounds, engine::WgPlan &P, int blocks, bool cut_ok, int lds_per_cu, int minv, int carry, int curv0, int curv_it, int inv_nb)
{
    return dispatch_model(m.model_id, [&](auto mdl) { return engine::wg_plan<decltype(mdl)>(m, hard, waves, state_bounds, P, blocks, cut_ok, lds_per_cu, minv, carry, curv0, curv_it, inv_nb); });
}

int nlmpc_wg_launch(const NlmpcDev *m, const NlmpcSolveDev *b, const engine::WgPlan *P, void *stream)
{
    return dispatch_model(m->model_id, [&](auto mdl) { return engine::launch_solve_wg<decltype(mdl)>(m, b, P, stream); });
}

}  // namespace mpcx
