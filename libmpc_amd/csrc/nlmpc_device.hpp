// POD views handed to the NLMPC transcription kernels (nlmpc_kernels.hip).
#pragma once

namespace mpcx {

struct NlmpcDev {
    int model_id, nx, nu, ph, ch, nz, neq, nineq;
    int lds_per_wave;           // doubles: X (ph+1)*nx | U (ph+1)*nu | scratch ph*nu
    double Ts;
    const double *params;       // model parameters in HBM
};

struct NlmpcBatchDev {
    int batch;
    const double *z;            // [B x nz]   decision vectors [X(1..ph) | U blocks | slack]
    const double *x0;           // [B x nx]
    double *cost, *grad;        // [B], [B x nz]
    double *ceq, *jeq;          // [B x ph*nx], [B x ph x nx x (2nx+nu)] blocks [dc/dx_i | dc/dx_{i+1} | dc/du_i]
    double *cineq, *jineq;      // [B x nineq], [B x nineq x nz] row-major
};

int nlmpc_model_dims(int model_id, int *nx, int *nu);
int nlmpc_launch(const NlmpcDev &m, const NlmpcBatchDev &b, void *stream);

}  // namespace mpcx
