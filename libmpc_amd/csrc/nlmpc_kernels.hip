// The NLMPC engine (mpcx/nlmpc_engine.hpp) instantiated for the model zoo of mpcx/nlmpc_models.hpp: the reference's example
// systems as component-wise device functors with declared constraint structure, selected by id (mpcx_nlmpc_create).
// Hooks with the reference's own signatures do not pass through this file: they instantiate the same engine in the user's
// translation unit (mpcx/nlmpc_hooks.hpp) or in a run-time compiled module (nlmpc_jit.cpp).
#include <hip/hip_runtime.h>

#include <atomic>
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
// Both are bound by the issue latency of dependent instructions; what a CU delivers is (instances resident) / (latency of one), the first set
// by the LDS block of the workgroup form.  The choice is a property of the CONTROLLER, not of the batch handed to one
// call: it is made once per handle and bound set (NlmpcZoo below) from the plan alone, so that a shard of a batch takes the kernel the
// whole batch would (the only thing a batch changes is where the folded blocks live -- LDS when the batch is resident with them there -- and
// that changes no arithmetic).  MPCX_NLMPC_FORM=wg|wave, MPCX_NLMPC_WAVES=1|2|4|8 and MPCX_NLMPC_BLOCKS=1|0 override; they are read when
// the handle is created, never on the solve path.
int nlmpc_wg_plan(const NlmpcDev &m, int hard, int waves, int state_bounds, engine::WgPlan &P, int blocks, bool cut_ok, int lds_per_cu, int minv, int carry, int curv0, int curv_it, int inv_nb);
int nlmpc_wg_launch(const NlmpcDev *m, const NlmpcSolveDev *b, const engine::WgPlan *P, void *stream);

static std::atomic<int> g_last_form{-1};
int nlmpc_last_form() { return g_last_form.load(std::memory_order_relaxed); }

// per-handle state of the launcher: the overrides, the device's limits, the plans of the current (hard / soft, bounds) shape
struct NlmpcZoo {
    int env_form = -1, env_waves = 0, env_blocks = -1, env_minv = -1, env_carry = -1, env_curv0 = -1, env_curv_it = -1, env_inv_nb = -1;          // -1 / 0: not set
    int lds_per_cu = 160 * 1024, cus = 256;
    int last_form = -1;
    NlmpcCurvLayout launched{};                               // where the last solve left its curvature estimate (form < 0: nowhere)
    // cache key and plans: throughput plan (most workgroups per CU; blocks where it puts them), the plan with the blocks in LDS, and the
    // plans of the second pass (full working-set capacity) for either
    int k_hard = -1, k_nbnd = -1, k_nbnd_state = -1, k_ws_total = -1;
    bool fits = false, has_lds = false, has_full = false, has_lds_full = false;
    engine::WgPlan P{}, Q{}, Pfull{}, Qfull{};
};

void *nlmpc_zoo_new()
{
    NlmpcZoo *z = new NlmpcZoo;
    z->launched.form = -1;
    const char *form = getenv("MPCX_NLMPC_FORM"), *wv = getenv("MPCX_NLMPC_WAVES"), *bl = getenv("MPCX_NLMPC_BLOCKS"),
               *mi = getenv("MPCX_NLMPC_MINV"), *ca = getenv("MPCX_NLMPC_CARRY"), *cv = getenv("MPCX_NLMPC_CURV0");
    if (cv) z->env_curv0 = atoi(cv) ? 1 : 0;
    if (const char *ci = getenv("MPCX_NLMPC_CURV0_IT")) z->env_curv_it = atoi(ci);
    if (const char *ib = getenv("MPCX_NLMPC_INV_NB")) z->env_inv_nb = atoi(ib);
    if (form) z->env_form = !strcmp(form, "wave") ? 0 : (!strcmp(form, "wg") ? 1 : -1);
    if (wv) z->env_waves = atoi(wv);
    if (bl) z->env_blocks = atoi(bl) ? 1 : 0;
    if (mi) z->env_minv = atoi(mi) ? 1 : 0;
    if (ca) z->env_carry = atoi(ca) ? 1 : 0;
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) == hipSuccess && v > 0) z->lds_per_cu = v;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) z->cus = v;
    }
    return z;
}
void nlmpc_zoo_free(void *z) { delete static_cast<NlmpcZoo *>(z); }
int nlmpc_zoo_last_form(void *z) { return z ? static_cast<NlmpcZoo *>(z)->last_form : -1; }

// the plans of the controller as it stands (hard / soft, bounds, workspace): made when one of those changed, not per solve
static void zoo_refresh(NlmpcZoo *z, const NlmpcDev *m, int hard)
{
    if (z->k_hard == hard && z->k_nbnd == m->nbnd && z->k_nbnd_state == m->nbnd_state && z->k_ws_total == m->ws.total) return;
    auto plan = [&](engine::WgPlan &X, int blocks, bool cut_ok, int waves = -1) {
        return nlmpc_wg_plan(*m, hard, waves < 0 ? z->env_waves : waves, m->nbnd_state, X, blocks, cut_ok, z->lds_per_cu, z->env_minv, z->env_carry, z->env_curv0, z->env_curv_it, z->env_inv_nb) == 0 && X.ws_total <= m->ws.total;
    };
    z->k_hard = hard; z->k_nbnd = m->nbnd; z->k_nbnd_state = m->nbnd_state; z->k_ws_total = m->ws.total;
    z->fits = plan(z->P, z->env_blocks, true);
    z->has_lds = z->fits && z->env_blocks < 0 && z->P.waves > 1 && !z->P.f_lds && plan(z->Q, 1, true) && z->Q.waves > 1;
    // (the second pass: the same wavefronts per instance -- the same arithmetic -- with the working set's full capacity)
    z->has_full = z->fits && plan(z->Pfull, z->P.f_lds, false, z->P.waves) && z->Pfull.kw > z->P.kw;
    z->has_lds_full = z->has_lds && plan(z->Qfull, 1, false, z->Q.waves) && z->Qfull.kw > z->Q.kw;
}
// the plan a batch of this size takes (nullptr: the wavefront form)
static const engine::WgPlan *zoo_pick(NlmpcZoo *z, int batch, bool &lds_plan)
{
    lds_plan = false;
    if (z->env_form == 0 || !z->fits) return nullptr;
    // the blocks in LDS where the whole batch is resident with them there (the latency form: config 3 at up to two instances per CU)
    lds_plan = z->has_lds && (long)batch <= (long)z->cus * z->Q.per_cu;
    return lds_plan ? &z->Q : &z->P;
}
static NlmpcCurvLayout layout_of(const engine::WgPlan *X, const NlmpcDev *m, int hard)
{
    NlmpcCurvLayout L{};
    L.form = X ? 1 : 0; L.waves = X ? X->waves : 0; L.f_lds = X ? X->f_lds : 0; L.w_hinv = X ? X->w_hinv : m->ws.hinv;
    L.hard = hard ? 1 : 0; L.nbnd_state = m->nbnd_state; L.nr = m->nr;
    return L;
}
// Carried curvature is valid only for a solve that reads the estimate where the previous one left it: the layout the NEXT launch of this
// batch will use (the plans are brought up to date first), and the one the LAST launch used
void nlmpc_zoo_next_layout(void *zp, const NlmpcDev *m, int hard, int batch, NlmpcCurvLayout *out)
{
    NlmpcZoo *z = static_cast<NlmpcZoo *>(zp);
    zoo_refresh(z, m, hard);
    bool lds_plan;
    *out = layout_of(zoo_pick(z, batch, lds_plan), m, hard);
}
void nlmpc_zoo_last_layout(void *zp, NlmpcCurvLayout *out) { *out = static_cast<NlmpcZoo *>(zp)->launched; }

int nlmpc_launch_solve(void *ctx, const NlmpcDev *m, const NlmpcSolveDev *b, void *stream)
{
    return dispatch_model(m->model_id, [&](auto mdl) {
        using Mdl = decltype(mdl);
        NlmpcZoo local;
        NlmpcZoo *z = ctx ? static_cast<NlmpcZoo *>(ctx) : &local;
        z->launched.form = -1;
        zoo_refresh(z, m, b->hard);
        bool lds_plan;
        if (const engine::WgPlan *Xp = zoo_pick(z, b->batch, lds_plan)) {
            const engine::WgPlan &X = *Xp;
            int rc = nlmpc_wg_launch(m, b, &X, stream);
            if (rc == 0) {
                // working sets that outgrew a capacity the plan had cut: those instances again, with the full one (the first pass has left their
                // start and their curvature estimate alone: WgSqp::finish)
                const bool again = lds_plan ? z->has_lds_full : z->has_full;
                if (again) {
                    engine::WgPlan Y = lds_plan ? z->Qfull : z->Pfull;
                    Y.only_overflowed = 1;
                    NlmpcSolveDev b2 = *b;
                    if (Y.w_hinv != X.w_hinv) b2.keep_curvature = 0;
                    rc = nlmpc_wg_launch(m, &b2, &Y, stream);
                }
                if (rc == 0) { z->last_form = X.waves; z->launched = layout_of(&X, m, b->hard); g_last_form.store(X.waves, std::memory_order_relaxed); return 0; }
            }
            if (z->env_form == 1) return rc;                // forced: report
            // otherwise (an attribute or launch error on this device): the wavefront form still runs
        } else if (z->env_form == 1) return -2;
        z->last_form = 0; g_last_form.store(0, std::memory_order_relaxed);
        NlmpcSolveDev bw = *b;
        if (z->env_form != 0 && z->fits) bw.keep_curvature = 0;      // (the caller counted on the workgroup form's layout)
        const int rc = engine::launch_solve<Mdl>(nullptr, m, &bw, stream);
        if (rc == 0) z->launched = layout_of(nullptr, m, b->hard);
        return rc;
    });
}

}  // namespace mpcx
