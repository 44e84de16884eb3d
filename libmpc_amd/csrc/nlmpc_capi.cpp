// extern "C" boundary of the NLMPC kernels (include/mpcx.h, mpcx_nlmpc_*).  Host code only; no CPU solve path.
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/mpcx.h"
#include "mpcx/nlmpc_device.hpp"

namespace mpcx {
int capi_fail(int code, const std::string &msg);
// zoo models (nlmpc_kernels.hip)
int nlmpc_model_dims(int model_id, int *nx, int *nu, int *ny, int ph, int *nineq, int *nue);
void nlmpc_plan_host(NlmpcDev &m);
int nlmpc_launch(void *, const NlmpcDev *m, const NlmpcBatchDev *b, void *stream);
int nlmpc_launch_solve(void *, const NlmpcDev *m, const NlmpcSolveDev *b, void *stream);
int nlmpc_last_form();
// the launcher's per-handle state for the built-in systems (nlmpc_kernels.hip): overrides read once, the device's limits, the cached plans
void *nlmpc_zoo_new();
void nlmpc_zoo_free(void *z);
int nlmpc_zoo_last_form(void *z);
void nlmpc_zoo_next_layout(void *z, const NlmpcDev *m, int hard, int batch, NlmpcCurvLayout *out);
void nlmpc_zoo_last_layout(void *z, NlmpcCurvLayout *out);
// run-time compiled hooks (nlmpc_jit.cpp)
void nlmpc_jit_release(void *jit);
}

struct mpcx_nlmpc {
    mpcx::NlmpcDev dev{};
    int device = 0;
    // how the two kernels are launched: the library's zoo, a user's translation unit, or a run-time compiled module
    mpcx::nlmpc_launch_eval_fn launch_eval = mpcx::nlmpc_launch;
    mpcx::nlmpc_launch_solve_fn launch_solve = mpcx::nlmpc_launch_solve;
    void *launch_ctx = nullptr;
    void *jit = nullptr;                 // run-time compiled module (nlmpc_jit.cpp), released with the handle
    void *zoo = nullptr;                 // built-in systems: the launcher's state (form, plans), = launch_ctx
    mpcx::NlmpcCurvLayout curv{-1, 0, 0, 0, 0, 0, 0};      // where the last solve left its curvature estimate in the workspace (form < 0: nowhere)
    double *params_d = nullptr;
    int n_params = 0;                   // parameters of the built-in system (0: none, or a hook model)
    double *scale_d = nullptr;           // input scaling [nu] | state scaling [nx] (Mapping.hpp:71-86), ones by default
    std::vector<double> su, ss;
    double *hook_ws = nullptr;           // scratch of mpcx_nlmpc_evaluate_batch for vector-valued hooks
    size_t hook_cap = 0;
    double *ws = nullptr;
    size_t ws_cap = 0;          // instances
    int solved_batch = 0;       // batch size of the last solve whose state is still in the workspace (0: none)
    mpcx_nlparams prm{};
    double tol_step = 1e-6, tol_con = 1e-8;      // own convergence test (mpcx_nlmpc_debug_set_tolerances: experiment knob)
    // NLOptimizer::lb / ub (NLOptimizer.hpp:346-404): bounds on the decision vector, host copy + device tables
    std::vector<double> lb, ub;
    bool bounds_dirty = true;
    void *bnd_block = nullptr;  // one allocation: zlb | zub | bnd_val | bnd_sign | bnd_idx

    int sync_scale()
    {
        std::vector<double> both(su);
        both.insert(both.end(), ss.begin(), ss.end());
        bool any = false;
        for (double v : both) any |= v != 1.0;
        for (double v : ss) both.push_back(1.0 / v);
        if (!scale_d && hipMalloc(reinterpret_cast<void **>(&scale_d), both.size() * sizeof(double)) != hipSuccess) return MPCX_E_DEVICE;
        if (hipMemcpy(scale_d, both.data(), both.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return MPCX_E_DEVICE;
        dev.su = scale_d; dev.ss = scale_d + su.size(); dev.iss = dev.ss + ss.size(); dev.scaled = any ? 1 : 0;
        return MPCX_OK;
    }
    int sync_bounds()
    {
        if (!bounds_dirty) return MPCX_OK;
        const int nz = dev.nz;
        std::vector<int> idx; std::vector<double> sign, val;
        for (int k = 0; k < nz - 1; ++k) {          // the slack is handled by hard_constraints
            if (ub[k] < 1e30) { idx.push_back(k); sign.push_back(1.0); val.push_back(ub[k]); }
            if (lb[k] > -1e30) { idx.push_back(k); sign.push_back(-1.0); val.push_back(lb[k]); }
        }
        const int nb = (int)idx.size();
        if (bnd_block) (void)hipFree(bnd_block);
        bnd_block = nullptr;
        const size_t bytes = sizeof(double) * (2 * (size_t)nz + 2 * (size_t)nb + 2) + sizeof(int) * ((size_t)nb + 2);
        if (hipMalloc(&bnd_block, bytes) != hipSuccess) return MPCX_E_DEVICE;
        double *d = static_cast<double *>(bnd_block);
        bool ok = hipMemcpy(d, lb.data(), sizeof(double) * nz, hipMemcpyHostToDevice) == hipSuccess &&
                  hipMemcpy(d + nz, ub.data(), sizeof(double) * nz, hipMemcpyHostToDevice) == hipSuccess;
        double *dval = d + 2 * nz, *dsign = dval + nb + 1;
        int *didx = reinterpret_cast<int *>(dsign + nb + 1);
        if (nb) {
            ok = ok && hipMemcpy(dval, val.data(), sizeof(double) * nb, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(dsign, sign.data(), sizeof(double) * nb, hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(didx, idx.data(), sizeof(int) * nb, hipMemcpyHostToDevice) == hipSuccess;
        }
        if (!ok) return MPCX_E_DEVICE;
        dev.zlb = d; dev.zub = d + nz; dev.bnd_val = dval; dev.bnd_sign = dsign; dev.bnd_idx = didx;
        const int nbs_before = dev.nbnd_state;
        dev.nbnd_state = 0;
        for (int k : idx) dev.nbnd_state += k < dev.ph * dev.nx ? 1 : 0;
        if (dev.nbnd_state != nbs_before) solved_batch = 0;      // (the workgroup form's layout depends on it: no carried curvature across the change)
        if (nb != dev.nbnd) {                       // the workspace layout depends on the number of rows
            dev.nbnd = nb;
            mpcx::nlmpc_plan_host(dev);
            if (ws) (void)hipFree(ws);
            ws = nullptr; ws_cap = 0; solved_batch = 0;
        }
        bounds_dirty = false;
        return MPCX_OK;
    }
};

extern "C" {

void mpcx_nlparams_default(mpcx_nlparams *p)
{
    if (!p) return;
    *p = mpcx_nlparams{100, 0.0, 0, -1.0, -1.0, -1.0, -1.0, 1};      // Types.hpp:108-143
}

// common tail of the three ways to create a controller: plan the workspace, default bounds and scalings
static int finish_create(mpcx_nlmpc *h, mpcx_nlmpc_t *out)
{
    using mpcx::capi_fail;
    mpcx::NlmpcDev &d = h->dev;
    d.nbnd = 0;
    mpcx::nlmpc_plan_host(d);
    h->lb.assign(d.nz, -INFINITY); h->ub.assign(d.nz, INFINITY);
    h->su.assign(d.nu, 1.0); h->ss.assign(d.nx, 1.0);
    mpcx_nlparams_default(&h->prm);
    if ((size_t)d.lds_per_wave * sizeof(double) > 64 * 1024) {
        mpcx_nlmpc_destroy(h);
        return capi_fail(MPCX_E_UNSUPPORTED, "horizon too long for the per-wave LDS slice");
    }
    if (h->sync_scale() != MPCX_OK) {
        mpcx_nlmpc_destroy(h);
        return capi_fail(MPCX_E_DEVICE, "could not upload the scalings");
    }
    if (h->launch_solve == mpcx::nlmpc_launch_solve && !h->launch_ctx) { h->zoo = mpcx::nlmpc_zoo_new(); h->launch_ctx = h->zoo; }
    *out = h;
    return MPCX_OK;
}

int mpcx_nlmpc_create(int model_id, int ph, int ch, double Ts, const double *params, int n_params, int device,
                      mpcx_nlmpc_t *out)
{
    using mpcx::capi_fail;
    if (!out) return capi_fail(MPCX_E_INVALID, "null output handle");
    if (ph < 1 || ch < 1 || ch > ph) return capi_fail(MPCX_E_INVALID, "need 1 <= ch <= ph");
    int nx = 0, nu = 0, ny = 0, nineq = 0, nue = 0;
    if (mpcx::nlmpc_model_dims(model_id, &nx, &nu, &ny, ph, &nineq, &nue) != 0) return capi_fail(MPCX_E_INVALID, "unknown NLMPC model id");
    std::vector<double> prm;
    int want = 0;
    if (model_id == MPCX_MODEL_UGV) { prm = {0.7071067811865476, 0.7071067811865476, 2.0, 1.0, 0.3, 1.0, 1.0, 0.3, Ts}; want = 9; }
    else if (model_id == MPCX_MODEL_OSCILLATORS6 || model_id == MPCX_MODEL_OSCILLATORS8) { prm = {1.0, 0.1}; want = 2; }
    else if (model_id == MPCX_MODEL_VANDERPOL_RATE) { prm = {0.1}; want = 1; }
    else prm = {0.0};
    if (params && n_params > 0) {
        if (n_params != want) return capi_fail(MPCX_E_INVALID, "this model takes " + std::to_string(want) + " parameters");
        prm.assign(params, params + want);
    }
    if (hipSetDevice(device) != hipSuccess) return capi_fail(MPCX_E_DEVICE, "hipSetDevice failed: no usable HIP device");
    auto *h = new mpcx_nlmpc;
    h->device = device;
    if (hipMalloc(reinterpret_cast<void **>(&h->params_d), prm.size() * sizeof(double)) != hipSuccess ||
        hipMemcpy(h->params_d, prm.data(), prm.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        mpcx_nlmpc_destroy(h);
        return capi_fail(MPCX_E_DEVICE, "could not upload the model parameters");
    }
    mpcx::NlmpcDev &d = h->dev;
    d.model_id = model_id; d.nx = nx; d.nu = nu; d.ph = ph; d.ch = ch; d.nineq = nineq; d.nue = nue; d.ny = ny;
    d.Ts = Ts;
    d.params = h->params_d;
    h->n_params = want;
    return finish_create(h, out);
}

// shared by mpcx_nlmpc_create_custom and the run-time compiled path (nlmpc_jit.cpp); internal: not exported
__attribute__((visibility("hidden"))) int mpcx_nlmpc_create_hooked(const mpcx_nlmpc_custom *c, double Ts, int device, void *jit, mpcx_nlmpc_t *out)
{
    // `jit` (the run-time compiled module, may be null) belongs to the handle from here on: every way out that creates no handle
    // releases it
    auto capi_fail = [&](int code, const char *msg) { if (jit) mpcx::nlmpc_jit_release(jit); return mpcx::capi_fail(code, msg); };
    if (!c || !out) return capi_fail(MPCX_E_INVALID, "null argument");
    if (c->nx < 1 || c->nu < 1 || c->ny < 0 || c->nineq < 0 || c->neq_user < 0) return capi_fail(MPCX_E_INVALID, "bad dimensions");
    if (c->ph < 1 || c->ch < 1 || c->ch > c->ph) return capi_fail(MPCX_E_INVALID, "need 1 <= ch <= ph");
    if (!c->launch_evaluate || !c->launch_solve) return capi_fail(MPCX_E_INVALID, "launch thunks are required");
    if (c->hooks_bytes < 0 || (c->hooks_bytes > 0 && !c->hooks)) return capi_fail(MPCX_E_INVALID, "bad hook blob");
    if (hipSetDevice(device) != hipSuccess) return capi_fail(MPCX_E_DEVICE, "hipSetDevice failed: no usable HIP device");
    auto *h = new mpcx_nlmpc;
    h->device = device;
    h->jit = jit;
    const size_t nb = c->hooks_bytes > 0 ? (size_t)c->hooks_bytes : sizeof(double);
    if (hipMalloc(reinterpret_cast<void **>(&h->params_d), nb) != hipSuccess ||
        (c->hooks_bytes > 0 && hipMemcpy(h->params_d, c->hooks, nb, hipMemcpyHostToDevice) != hipSuccess)) {
        mpcx_nlmpc_destroy(h);                       // releases the module too
        return mpcx::capi_fail(MPCX_E_DEVICE, "could not upload the hook closures");
    }
    h->launch_eval = reinterpret_cast<mpcx::nlmpc_launch_eval_fn>(c->launch_evaluate);
    h->launch_solve = reinterpret_cast<mpcx::nlmpc_launch_solve_fn>(c->launch_solve);
    h->launch_ctx = c->launch_ctx;
    mpcx::NlmpcDev &d = h->dev;
    d.model_id = 0; d.nx = c->nx; d.nu = c->nu; d.ny = c->ny; d.ph = c->ph; d.ch = c->ch; d.nineq = c->nineq; d.nue = c->neq_user;
    d.Ts = Ts; d.continuous = Ts > 0.0 ? 1 : 0; d.has_output = c->has_output ? 1 : 0; d.vector_hooks = c->vector_hooks ? 1 : 0;
    d.params = h->params_d;
    return finish_create(h, out);
}

int mpcx_nlmpc_create_custom(const mpcx_nlmpc_custom *c, double Ts, int device, mpcx_nlmpc_t *out)
{
    return mpcx_nlmpc_create_hooked(c, Ts, device, nullptr, out);
}

int mpcx_nlmpc_destroy(mpcx_nlmpc_t h)
{
    if (!h) return MPCX_OK;
    (void)hipSetDevice(h->device);
    if (h->params_d) (void)hipFree(h->params_d);
    if (h->scale_d) (void)hipFree(h->scale_d);
    if (h->hook_ws) (void)hipFree(h->hook_ws);
    if (h->jit) mpcx::nlmpc_jit_release(h->jit);
    if (h->zoo) mpcx::nlmpc_zoo_free(h->zoo);
    if (h->ws) (void)hipFree(h->ws);
    if (h->bnd_block) (void)hipFree(h->bnd_block);
    delete h;
    return MPCX_OK;
}

int mpcx_nlmpc_get_dims(mpcx_nlmpc_t h, mpcx_nlmpc_dims *d)
{
    if (!h || !d) return mpcx::capi_fail(MPCX_E_INVALID, "null argument");
    const mpcx::NlmpcDev &m = h->dev;
    int nb = 0;                                   // finite bounds = rows of the sub-problem after the user constraints
    for (int k = 0; k < m.nz - 1; ++k) nb += (h->ub[k] < 1e30) + (h->lb[k] > -1e30);
    *d = mpcx_nlmpc_dims{m.nx, m.nu, m.ph, m.ch, m.nz, m.neq, m.nineq, 2 * m.nx + m.nu, m.nue, m.ny, nb, h->n_params};
    return MPCX_OK;
}

int mpcx_nlmpc_set_optimizer_parameters(mpcx_nlmpc_t h, const mpcx_nlparams *p)
{
    if (!h || !p) return mpcx::capi_fail(MPCX_E_INVALID, "null argument");
    if (p->maximum_iteration < 0) return mpcx::capi_fail(MPCX_E_INVALID, "maximum_iteration must be >= 0");
    h->prm = *p;
    return MPCX_OK;
}

static int set_bounds(mpcx_nlmpc_t h, const double *lo, const double *hi, int start, int end, bool state)
{
    using mpcx::capi_fail;
    if (!h || !lo || !hi) return capi_fail(MPCX_E_INVALID, "null argument");
    const mpcx::NlmpcDev &d = h->dev;
    const int horizon = state ? d.ph : d.ch, n = state ? d.nx : d.nu, base = state ? 0 : d.ph * d.nx;
    // HorizonSlice validity as IMPC.hpp:244-283: {-1,-1} = everything, otherwise 0 <= start < end <= horizon
    const bool unset = start == -1 && end == -1;
    if (!unset && !(start >= 0 && end > start && end <= horizon)) return capi_fail(MPCX_E_INVALID, "invalid horizon slice");
    const int a = unset ? 0 : start, b = unset ? horizon : end;
    for (int j = 0; j < n; ++j) if (lo[j] > hi[j]) return capi_fail(MPCX_E_INVALID, "lower bound above upper bound");
    for (int i = a; i < b; ++i)
        for (int j = 0; j < n; ++j) { h->lb[base + i * n + j] = lo[j]; h->ub[base + i * n + j] = hi[j]; }
    h->bounds_dirty = true;
    return MPCX_OK;
}

int mpcx_nlmpc_set_state_bounds_slice(mpcx_nlmpc_t h, const double *lo, const double *hi, int start, int end)
{
    return set_bounds(h, lo, hi, start, end, true);
}

int mpcx_nlmpc_set_input_bounds_slice(mpcx_nlmpc_t h, const double *lo, const double *hi, int start, int end)
{
    return set_bounds(h, lo, hi, start, end, false);
}

static int set_scale(mpcx_nlmpc_t h, const double *s, bool state)
{
    using mpcx::capi_fail;
    if (!h || !s) return capi_fail(MPCX_E_INVALID, "null argument");
    std::vector<double> &dst = state ? h->ss : h->su;
    for (size_t j = 0; j < dst.size(); ++j) if (!(s[j] != 0.0) || !std::isfinite(s[j])) return capi_fail(MPCX_E_INVALID, "scaling factors must be finite and non-zero");
    dst.assign(s, s + dst.size());
    if (hipSetDevice(h->device) != hipSuccess || h->sync_scale() != MPCX_OK) return capi_fail(MPCX_E_DEVICE, "could not upload the scalings");
    h->solved_batch = 0;
    return MPCX_OK;
}
int mpcx_nlmpc_set_input_scale(mpcx_nlmpc_t h, const double *scaling) { return set_scale(h, scaling, false); }
int mpcx_nlmpc_set_state_scale(mpcx_nlmpc_t h, const double *scaling) { return set_scale(h, scaling, true); }

int mpcx_nlmpc_evaluate_batch(mpcx_nlmpc_t h, int batch, const double *z, const double *x0, double *cost, double *grad,
                              double *ceq, double *jeq, double *cineq, double *jineq, void *stream)
{
    using mpcx::capi_fail;
    if (!h) return capi_fail(MPCX_E_INVALID, "null handle");
    if (batch < 0) return capi_fail(MPCX_E_INVALID, "negative batch");
    if (batch == 0) return MPCX_OK;
    if (!z || !x0) return capi_fail(MPCX_E_INVALID, "z and x0 are required");
    if (hipSetDevice(h->device) != hipSuccess) return capi_fail(MPCX_E_DEVICE, "hipSetDevice failed");
    mpcx::NlmpcBatchDev b{batch, z, x0, cost, grad, ceq, jeq, cineq, jineq, nullptr, 0, nullptr, 0};
    if (h->dev.vector_hooks) {
        const int ld = mpcx::nlmpc_hook_scratch(h->dev);
        if ((size_t)batch > h->hook_cap) {
            if (h->hook_ws) (void)hipFree(h->hook_ws);
            h->hook_ws = nullptr; h->hook_cap = 0;
            if (hipMalloc(reinterpret_cast<void **>(&h->hook_ws), (size_t)batch * ld * sizeof(double)) != hipSuccess)
                return capi_fail(MPCX_E_DEVICE, "could not allocate the hook scratch");
            h->hook_cap = batch;
        }
        b.hook_ws = h->hook_ws; b.hook_ld = ld;
    }
    const int rc = h->launch_eval(h->launch_ctx, &h->dev, &b, stream);
    if (rc != 0) return capi_fail(MPCX_E_DEVICE, "NLMPC kernel launch failed (" + std::to_string(rc) + ")");
    return MPCX_OK;
}

static int prepare_solve(mpcx_nlmpc_t h, const mpcx_nlmpc_batch *b, mpcx::NlmpcSolveDev &s)
{
    using mpcx::capi_fail;
    if (!h || !b) return capi_fail(MPCX_E_INVALID, "null argument");
    if (b->batch < 0) return capi_fail(MPCX_E_INVALID, "negative batch");
    if (b->batch == 0) return 1;
    if (!b->x0 || !b->u0 || !b->cmd) return capi_fail(MPCX_E_INVALID, "x0, u0 and cmd are required");
    if (hipSetDevice(h->device) != hipSuccess) return capi_fail(MPCX_E_DEVICE, "hipSetDevice failed");
    if (h->sync_bounds() != MPCX_OK) return capi_fail(MPCX_E_DEVICE, "could not upload the bounds");
    if ((size_t)b->batch > h->ws_cap) {
        if (h->ws) (void)hipFree(h->ws);
        h->ws = nullptr; h->ws_cap = 0; h->solved_batch = 0;
        if (hipMalloc(reinterpret_cast<void **>(&h->ws), (size_t)b->batch * h->dev.ws.total * sizeof(double)) != hipSuccess)
            return capi_fail(MPCX_E_DEVICE, "could not allocate the SQP workspace");
        h->ws_cap = b->batch;
    }
    s = mpcx::NlmpcSolveDev{};
    s.batch = b->batch; s.x0 = b->x0; s.u0 = b->u0; s.z_warm = b->z_warm; s.ws = h->ws;
    s.max_iter = h->prm.maximum_iteration; s.hard = h->prm.hard_constraints ? 1 : 0;
    s.tol_step = h->tol_step;
    s.tol_con = h->tol_con; s.ieq_tol = 1e-10; s.eq_tol = 1e-10;
    s.ftol_rel = h->prm.relative_ftol; s.ftol_abs = h->prm.absolute_ftol;        // NLOptimizer.hpp:135-138, <= 0: disabled
    s.xtol_rel = h->prm.relative_xtol; s.xtol_abs = h->prm.absolute_xtol;
    s.mu_out = b->multipliers;
    if (b->params) {
        if (h->n_params <= 0) return capi_fail(MPCX_E_INVALID, "this model has no parameters to give per instance");
        s.params_b = b->params; s.nparams = h->n_params;
    }
    // (z_warm may be the same buffer as z: an instance reads its start before it writes its result, and one that a second launch takes again
    // -- a working set that outgrew a cut capacity -- has neither its start nor its curvature estimate overwritten by the first: WgSqp::finish)
    // carried curvature: only from a solve of the same batch that left it in the layout the next launch will read (form, wavefronts, offsets)
    mpcx::NlmpcCurvLayout next{0, 0, 0, h->dev.ws.hinv, s.hard, h->dev.nbnd_state, h->dev.nr};      // (hook models: always nlmpc_sqp)
    if (h->zoo) mpcx::nlmpc_zoo_next_layout(h->zoo, &h->dev, s.hard, b->batch, &next);
    s.keep_curvature = (b->warm_curvature && b->z_warm && h->solved_batch == b->batch && next == h->curv) ? 1 : 0;
    h->solved_batch = b->batch;
    s.cmd = b->cmd; s.cost = b->cost; s.z_out = b->z; s.status = b->status; s.solver_status = b->solver_status;
    s.is_feasible = b->is_feasible; s.iterations = b->iterations; s.seq_state = b->seq_state; s.seq_input = b->seq_input; s.seq_output = b->seq_output;
    return MPCX_OK;
}

// after a launch: where this solve's curvature estimate lies
static void note_curvature(mpcx_nlmpc_t h, const mpcx::NlmpcSolveDev &s)
{
    h->curv = mpcx::NlmpcCurvLayout{0, 0, 0, h->dev.ws.hinv, s.hard, h->dev.nbnd_state, h->dev.nr};
    if (h->zoo) mpcx::nlmpc_zoo_last_layout(h->zoo, &h->curv);
}

int mpcx_nlmpc_solve_batch(mpcx_nlmpc_t h, const mpcx_nlmpc_batch *b, void *stream)
{
    mpcx::NlmpcSolveDev s;
    const int rc = prepare_solve(h, b, s);
    if (rc != MPCX_OK) return rc > 0 ? MPCX_OK : rc;
    const int lr = h->launch_solve(h->launch_ctx, &h->dev, &s, stream);
    if (lr != 0) { h->solved_batch = 0; h->curv.form = -1; return mpcx::capi_fail(MPCX_E_DEVICE, "NLMPC solve launch failed (" + std::to_string(lr) + ")"); }
    note_curvature(h, s);
    return MPCX_OK;
}

int mpcx_nlmpc_solve_host(mpcx_nlmpc_t h, int batch, const double *x0, const double *u0, const double *z_warm, double *cmd,
                          double *cost, int32_t *status, int32_t *solver_status, int32_t *is_feasible, int32_t *iterations,
                          double *z, double *seq_state, double *seq_input)
{
    using mpcx::capi_fail;
    if (!h) return capi_fail(MPCX_E_INVALID, "null handle");
    if (batch < 0) return capi_fail(MPCX_E_INVALID, "negative batch");
    if (batch == 0) return MPCX_OK;
    if (!x0 || !u0 || !cmd) return capi_fail(MPCX_E_INVALID, "x0, u0 and cmd are required");
    if (hipSetDevice(h->device) != hipSuccess) return capi_fail(MPCX_E_DEVICE, "hipSetDevice failed");
    const mpcx::NlmpcDev &d = h->dev;
    const size_t B = (size_t)batch, n1 = (size_t)d.ph + 1;
    const size_t nd = B * (d.nx + d.nu + 2 * (size_t)d.nz + d.nu + 1 + n1 * (d.nx + d.nu));
    double *dbuf = nullptr; int32_t *ibuf = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&dbuf), nd * sizeof(double)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&ibuf), B * 4 * sizeof(int32_t)) != hipSuccess) {
        if (dbuf) (void)hipFree(dbuf);
        return capi_fail(MPCX_E_DEVICE, "staging allocation failed");
    }
    double *dx0 = dbuf, *du0 = dx0 + B * d.nx, *dzw = du0 + B * d.nu, *dz = dzw + B * d.nz, *dcmd = dz + B * d.nz,
           *dcost = dcmd + B * d.nu, *dss = dcost + B, *dsi = dss + B * n1 * d.nx;
    bool ok = hipMemcpy(dx0, x0, B * d.nx * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(du0, u0, B * d.nu * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    if (z_warm) ok = ok && hipMemcpy(dzw, z_warm, B * d.nz * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    mpcx_nlmpc_batch b{};
    b.batch = batch; b.x0 = dx0; b.u0 = du0; b.z_warm = z_warm ? dzw : nullptr; b.cmd = dcmd; b.cost = dcost;
    b.status = ibuf; b.solver_status = ibuf + B; b.is_feasible = ibuf + 2 * B; b.iterations = ibuf + 3 * B;
    b.z = dz; b.seq_state = dss; b.seq_input = dsi;
    int rc = ok ? mpcx_nlmpc_solve_batch(h, &b, nullptr) : MPCX_E_DEVICE;
    if (rc == MPCX_OK) {
        auto back = [&](void *dst, const void *src, size_t bytes) { if (dst) ok = ok && hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess; };
        back(cmd, dcmd, B * d.nu * sizeof(double)); back(cost, dcost, B * sizeof(double));
        back(status, ibuf, B * 4); back(solver_status, ibuf + B, B * 4); back(is_feasible, ibuf + 2 * B, B * 4);
        back(iterations, ibuf + 3 * B, B * 4); back(z, dz, B * d.nz * sizeof(double));
        back(seq_state, dss, B * n1 * d.nx * sizeof(double)); back(seq_input, dsi, B * n1 * d.nu * sizeof(double));
        if (!ok) rc = capi_fail(MPCX_E_DEVICE, "copy failed");
    } else if (!ok) {
        rc = capi_fail(MPCX_E_DEVICE, "copy failed");
    }
    (void)hipFree(dbuf); (void)hipFree(ibuf);
    return rc;
}

int mpcx_nlmpc_time_solve_batch(mpcx_nlmpc_t h, const mpcx_nlmpc_batch *b, void *stream, int repeats, float *ms_mean)
{
    using mpcx::capi_fail;
    if (!ms_mean || repeats < 1) return capi_fail(MPCX_E_INVALID, "bad timing arguments");
    mpcx::NlmpcSolveDev s;
    const int rc = prepare_solve(h, b, s);
    if (rc != MPCX_OK) { *ms_mean = 0.f; return rc > 0 ? MPCX_OK : rc; }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return capi_fail(MPCX_E_DEVICE, "hipEventCreate failed");
    (void)hipEventRecord(e0, st);
    int lr = 0;
    for (int i = 0; i < repeats && lr == 0; ++i) lr = h->launch_solve(h->launch_ctx, &h->dev, &s, stream);
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (lr != 0) { h->solved_batch = 0; h->curv.form = -1; return capi_fail(MPCX_E_DEVICE, "NLMPC solve launch failed"); }
    note_curvature(h, s);
    *ms_mean = ms / repeats;
    return MPCX_OK;
}

}  // extern "C"

namespace mpcx {
int c2d_launch(int nx, int nu, int batch, const double *A, const double *B, const double *Ts, int ts_stride, double *Ad, double *Bd,
               void *stream);
}

extern "C" int mpcx_discretize_batch(int device, int nx, int nu, int batch, const double *A, const double *B, const double *Ts,
                                     int ts_per_instance, double *Ad, double *Bd, void *stream)
{
    using mpcx::capi_fail;
    if (nx < 1 || nu < 0 || batch < 0) return capi_fail(MPCX_E_INVALID, "bad dimensions");
    if (batch == 0) return MPCX_OK;
    if (!A || (nu > 0 && !B) || !Ts || !Ad || (nu > 0 && !Bd)) return capi_fail(MPCX_E_INVALID, "null argument");
    if (nx + nu > 48) return capi_fail(MPCX_E_UNSUPPORTED, "nx + nu > 48");
    if (hipSetDevice(device) != hipSuccess) return capi_fail(MPCX_E_DEVICE, "hipSetDevice failed");
    const int rc = mpcx::c2d_launch(nx, nu, batch, A, B, Ts, ts_per_instance ? 1 : 0, Ad, Bd, stream);
    if (rc != 0) return capi_fail(MPCX_E_DEVICE, "discretisation kernel launch failed");
    return MPCX_OK;
}

// Which kernel the last solve of a built-in system went through (bench.py names it): 0 = nlmpc_sqp (one wavefront per instance),
// 1 | 2 | 4 = nlmpc_sqp_wg with that many wavefronts per instance, -1 = none yet
extern "C" int mpcx_nlmpc_debug_last_form(void) { return mpcx::nlmpc_last_form(); }
// ... and the last solve of THIS handle (the one above is the last launch of the process, whoever made it)
extern "C" int mpcx_nlmpc_last_form(mpcx_nlmpc_t h) { return (h && h->zoo) ? mpcx::nlmpc_zoo_last_form(h->zoo) : (h ? 0 : -1); }

// Experiment knob (not part of include/mpcx.h): the step / defect thresholds of the solver's own convergence test
extern "C" int mpcx_nlmpc_debug_set_tolerances(mpcx_nlmpc_t h, double tol_step, double tol_con)
{
    if (!h || !(tol_step > 0) || !(tol_con > 0)) return mpcx::capi_fail(MPCX_E_INVALID, "bad tolerances");
    h->tol_step = tol_step; h->tol_con = tol_con;
    return MPCX_OK;
}

// Testing aid (not part of include/mpcx.h): copy one instance's SQP workspace to the host together with its layout.
extern "C" int mpcx_nlmpc_debug_get_ws(mpcx_nlmpc_t h, int instance, double *out, int cap, int *layout, int nlayout)
{
    if (!h || !h->ws || instance < 0 || (size_t)instance >= h->ws_cap) return mpcx::capi_fail(MPCX_E_INVALID, "no such workspace");
    const int total = h->dev.ws.total;
    if (layout) {
        const int *src = reinterpret_cast<const int *>(&h->dev.ws);
        const int n = (int)(sizeof(mpcx::NlmpcWsLayout) / sizeof(int));
        for (int i = 0; i < n && i < nlayout; ++i) layout[i] = src[i];
    }
    if (!out) return total;
    if (cap < total) return mpcx::capi_fail(MPCX_E_INVALID, "buffer too small");
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    if (hipMemcpy(out, h->ws + (size_t)instance * total, (size_t)total * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        return mpcx::capi_fail(MPCX_E_DEVICE, "copy failed");
    return total;
}
