// TEST INFRASTRUCTURE (host interpreter only): probe points of include/mpcx/nlmpc_sqp_wg.hpp.  Included BEFORE that header by the
// emulator runner when it is built with -DHIPEMU_CHECK_CARRY: the carried inverse of the working set's Schur complement (ws_warm) against
// the Schur complement formed afresh, |S M - I| per carried warm start on stderr.
#pragma once
#include <cmath>
#include <cstdio>

namespace hipemu_probe {
template <class SP>
inline void check_carry(bool have_m, int nw, const int *wq, const double *sgq, const SP &sp, const double *hinv, const double *Lp, bool upd, double carried)
{
    if (threadIdx.x != 0 || !have_m) return;
    auto sym = [](const double *hp, int r, int c) { return r >= c ? hp[r * (r + 1) / 2 + c] : hp[c * (c + 1) / 2 + r]; };
    double worst = 0.0;
    for (int a = 0; a < nw; ++a)
        for (int b2 = 0; b2 < nw; ++b2) {
            double acc = a == b2 ? -1.0 : 0.0;
            for (int c = 0; c < nw; ++c) {
                const int ka = wq[a], kc = wq[c];
                double sac = 0.0;
                for (int ja = 0; ja < sp.count(ka); ++ja)
                    for (int jc = 0; jc < sp.count(kc); ++jc) sac = std::fma(sp.value(ka, ja) * sp.value(kc, jc), sym(hinv, sp.index(ka, ja), sp.index(kc, jc)), sac);
                acc += sgq[a] * sgq[c] * sac * sym(Lp, c, b2);
            }
            worst = std::fmax(worst, std::fabs(acc));
        }
    fprintf(stderr, "carry check: nw %d upd %d carried %g  |S M - I| = %.3e\n", nw, (int)upd, carried, worst);
}
}  // namespace hipemu_probe
#define MPCX_WG_PROBE_CARRY(...) hipemu_probe::check_carry(__VA_ARGS__)
