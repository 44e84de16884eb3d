// extern "C" boundary of the NLMPC transcription kernels (include/mpcx.h, mpcx_nlmpc_*).
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/mpcx.h"
#include "nlmpc_device.hpp"

namespace mpcx {
int capi_fail(int code, const std::string &msg);
}

struct mpcx_nlmpc {
    mpcx::NlmpcDev dev{};
    int device = 0;
    double *params_d = nullptr;
};

extern "C" {

int mpcx_nlmpc_create(int model_id, int ph, int ch, double Ts, const double *params, int n_params, int device,
                      mpcx_nlmpc_t *out)
{
    using mpcx::capi_fail;
    if (!out) return capi_fail(MPCX_E_INVALID, "null output handle");
    int nx = 0, nu = 0;
    if (mpcx::nlmpc_model_dims(model_id, &nx, &nu) != 0) return capi_fail(MPCX_E_INVALID, "unknown NLMPC model id");
    if (ph < 1 || ch < 1 || ch > ph) return capi_fail(MPCX_E_INVALID, "need 1 <= ch <= ph");
    std::vector<double> prm;
    if (model_id == MPCX_MODEL_UGV) {
        prm = {0.7071067811865476, 0.7071067811865476, 2.0, 1.0, 0.3, 1.0, 1.0, 0.3, Ts};
        if (params) {
            if (n_params != 9) return capi_fail(MPCX_E_INVALID, "the UGV model takes 9 parameters");
            prm.assign(params, params + 9);
        }
    } else {
        prm = {0.0};
        if (params && n_params != 0) return capi_fail(MPCX_E_INVALID, "the Van der Pol model takes no parameters");
    }
    if (hipSetDevice(device) != hipSuccess) return capi_fail(MPCX_E_DEVICE, "hipSetDevice failed: no usable HIP device");
    auto *h = new mpcx_nlmpc;
    h->device = device;
    if (hipMalloc(reinterpret_cast<void **>(&h->params_d), prm.size() * sizeof(double)) != hipSuccess ||
        hipMemcpy(h->params_d, prm.data(), prm.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) {
        delete h;
        return capi_fail(MPCX_E_DEVICE, "could not upload the model parameters");
    }
    mpcx::NlmpcDev &d = h->dev;
    d.model_id = model_id; d.nx = nx; d.nu = nu; d.ph = ph; d.ch = ch;
    d.nz = ph * nx + ch * nu + 1;
    d.neq = ph * nx;
    d.nineq = model_id == MPCX_MODEL_UGV ? 2 * (ph + 1) : ph + 1;
    d.lds_per_wave = ((ph + 1) * (nx + nu) + ph * nu + 1) & ~1;
    d.Ts = Ts;
    d.params = h->params_d;
    *out = h;
    return MPCX_OK;
}

int mpcx_nlmpc_destroy(mpcx_nlmpc_t h)
{
    if (!h) return MPCX_OK;
    (void)hipSetDevice(h->device);
    if (h->params_d) (void)hipFree(h->params_d);
    delete h;
    return MPCX_OK;
}

int mpcx_nlmpc_get_dims(mpcx_nlmpc_t h, mpcx_nlmpc_dims *d)
{
    if (!h || !d) return mpcx::capi_fail(MPCX_E_INVALID, "null argument");
    const mpcx::NlmpcDev &m = h->dev;
    *d = mpcx_nlmpc_dims{m.nx, m.nu, m.ph, m.ch, m.nz, m.neq, m.nineq, 2 * m.nx + m.nu};
    return MPCX_OK;
}

int mpcx_nlmpc_evaluate_batch(mpcx_nlmpc_t h, int batch, const double *z, const double *x0, double *cost, double *grad,
                              double *ceq, double *jeq, double *cineq, double *jineq, void *stream)
{
    using mpcx::capi_fail;
    if (!h) return capi_fail(MPCX_E_INVALID, "null handle");
    if (batch < 0) return capi_fail(MPCX_E_INVALID, "negative batch");
    if (batch == 0) return MPCX_OK;
    if (!z || !x0) return capi_fail(MPCX_E_INVALID, "z and x0 are required");
    if (hipSetDevice(h->device) != hipSuccess) return capi_fail(MPCX_E_DEVICE, "hipSetDevice failed");
    mpcx::NlmpcBatchDev b{batch, z, x0, cost, grad, ceq, jeq, cineq, jineq};
    const int rc = mpcx::nlmpc_launch(h->dev, b, stream);
    if (rc != 0) return capi_fail(MPCX_E_DEVICE, "NLMPC kernel launch failed (" + std::to_string(rc) + ")");
    return MPCX_OK;
}

}  // extern "C"
