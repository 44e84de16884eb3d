// extern "C" boundary of the batched LMPC engine (include/mpcx.h).
// Host code only: owns the controller state, runs the one-time condensing, keeps the
// condensed model resident in HBM, and launches the HIP kernel.  There is no CPU solve
// path here: without a usable HIP device every solve call fails with MPCX_E_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <memory>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/mpcx.h"
#include "lmpc_device.hpp"
#include "lmpc_model.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

}  // namespace

namespace mpcx {
int capi_fail(int code, const std::string &msg) { return fail(code, msg); }   // shared with nlmpc_capi.cpp
}

struct mpcx_lmpc {
    mpcx::LmpcController ctl;
    int device = 0;
    bool host_only = false;
    bool dirty = true;                  // controller changed: condense again, rebuild the device-resident model
    bool refs_dirty = false;            // only references / exogenous inputs changed: refresh what depends on them, in place
    mpcx::Condensed cond;
    mpcx::LmpcDev dev{};
    mpcx::LmpcDev *dev_d = nullptr;     // the same struct, resident in HBM for the kernel
    std::vector<void *> allocs;
    long long *dbg_cycles = nullptr;
    bool force_generic = false;
    bool strict_infeasible = false;
    int dbg_rounds0 = 30, dbg_check_every = 10;      // experiment knobs (mpcx_lmpc_debug_set_rounds)
    // Fused forms (lmpc_solve_fused / lmpc_solve_persistent): the instance's record is computed inside the solve kernel instead of
    // being handed over through the HBM workspace (3 MB instead of 25 MB of traffic per 4096 instances), but the hardest-first
    // dispatch order of the two-kernel path is lost -- worth 22 us of its 70 us at 4096 instances, nothing at large batches.
    // Measured (quadrotor N = 20, ms per step, two kernels / fused): 4096: 0.097 / 0.125; 16384: 0.309 / 0.300; 65536: 1.10 / 1.00.
    // -1 = automatic (lmpc_solve_group up to group_max instances, assemble + solve as two kernels beyond; the fused / persistent mat-vec forms
    // only on request), 0 = always two kernels, 1 = the fused form wherever the dimensions allow, 2 = the group form (mpcx_lmpc_debug_use_fused)
    int use_fused = -1;
    int group_max = 4096;               // automatic mode: batches up to this size take lmpc_solve_group (assemble + solve in one workgroup)
    int total_batch = 0;                // mpcx_lmpc_set_total_batch: this handle solves shards of a batch of that size (0: every call is a whole batch)
    // staging of mpcx_lmpc_solve_host (kept between calls) and the active sets it carries from one call to the next
    double *stage_d = nullptr; int32_t *stage_i = nullptr; uint32_t *stage_act = nullptr;
    size_t stage_cap = 0;               // instances
    int warm_batch = 0;                 // batch size whose active sets are in stage_act (0: none)
    int n_full_setups = 0, n_ref_refreshes = 0;      // how often each kind of set-up ran (mpcx_lmpc_debug_setup_counts)
    double *ws = nullptr;               // per-instance workspace between assemble and solve
    int *pcounter = nullptr;            // work counters of the persistent fused kernel (eight ints of its own)
    int32_t *done = nullptr;            // [ws_cap] lmpc_solve_group: which instances it solved (the fallback kernel screens this instead of the records)
    size_t ws_cap = 0;                  // instances
    explicit mpcx_lmpc(const mpcx_dims &d) : ctl(d) {}

    void release()
    {
        for (void *p : allocs) (void)hipFree(p);
        allocs.clear();
        if (ws) (void)hipFree(ws);
        if (pcounter) (void)hipFree(pcounter);
        if (done) (void)hipFree(done);
        ws = nullptr; pcounter = nullptr; done = nullptr; ws_cap = 0;
        warm_batch = 0;                 // row numbering may have changed with the model
    }
    void release_staging()
    {
        if (stage_d) (void)hipFree(stage_d);
        if (stage_i) (void)hipFree(stage_i);
        if (stage_act) (void)hipFree(stage_act);
        stage_d = nullptr; stage_i = nullptr; stage_act = nullptr; stage_cap = 0; warm_batch = 0;
    }
    // copy into a buffer this handle already owns on the device (same size as at set-up)
    template <typename T>
    bool reup(const T *dst, const std::vector<T> &v)
    {
        return v.empty() || hipMemcpy(const_cast<T *>(dst), v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
    }
    // the O(n^3) arrays (H, Kinv, Gr, Gc, Y): for a single controller they are uploaded like everything else
    template <typename T> const T *up_big(const std::vector<T> &v, int &rc) { return up(v, rc); }
    const double *reserve_big(size_t) { return nullptr; }         // (banks only)
    template <typename T>
    const T *up(const std::vector<T> &v, int &rc)
    {
        size_t n = v.size() ? v.size() : 1;
        void *p = nullptr;
        if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) { rc = MPCX_E_DEVICE; return nullptr; }
        allocs.push_back(p);
        if (v.size() && hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) rc = MPCX_E_DEVICE;
        return static_cast<const T *>(p);
    }
};

// The device view of one condensed controller: scalars ...
static void fill_dev_scalars(const mpcx_lmpc *h, const mpcx::LmpcController &c, const mpcx::Condensed &o, mpcx::LmpcDev &D)
{
    D = mpcx::LmpcDev{};
    D.nx = c.d.nx; D.nu = c.d.nu; D.ndu = c.d.ndu; D.ny = c.d.ny; D.ph = c.d.ph; D.ch = c.d.ch;
    D.nf = o.nf; D.nz = o.nz; D.mg = o.mg; D.ldz = o.ldz; D.ldg = o.ldg; D.ldy = o.ldy;
    D.m_ref = o.m_ref; D.neq_ref = o.neq_ref; D.active_words = o.active_words;
    D.has_dist = o.has_dist ? 1 : 0;
    D.n_fixed = (int)o.fixed_rows.size();
    D.max_iter = c.prm.maximum_iteration; D.polish = c.prm.polish ? 1 : 0;
    D.strict_infeasible = h->strict_infeasible ? 1 : 0;
    D.cost_direct = (o.h_regularised || o.inverse_residual > 1e-9) ? 1 : 0;
    D.check_every = h->dbg_check_every; D.polish_rounds0 = h->dbg_rounds0; D.polish_rounds = 10;
    D.alpha = c.prm.alpha; D.sigma = 1e-6;
    D.adaptive_rho = c.prm.adaptive_rho ? 1 : 0; D.rho_user = c.prm.rho;
    D.eps_abs = c.prm.eps_abs; D.eps_rel = c.prm.eps_rel; D.eps_prim_inf = c.prm.eps_prim_inf;
    D.lds_per_wave = mpcx::lmpc_lds_per_wave(D, &D.stage_len, &D.arena_len);
    D.wsld = D.ldz + D.ldy + 2 * D.ldg + 2;
    D.s0lo = c.sMin[0]; D.s0hi = c.sMax[0];
}

// ... and arrays, through an uploader (one hipMalloc per array for a single controller, offsets into one slab for a bank of them)
template <class Uploader>
static void fill_dev_arrays(const mpcx_lmpc *h, const mpcx::LmpcController &c, const mpcx::Condensed &o, mpcx::LmpcDev &D, Uploader &U, int &rc)
{
    D.A = U.up(c.A.a, rc); D.B = U.up(c.B.a, rc); D.C = U.up(c.C.a, rc);
    D.Bd = U.up(c.Bd.a, rc); D.Dd = U.up(c.Dd.a, rc);
    D.Wy = U.up(c.wOutput.a, rc); D.Wu = U.up(c.wU.a, rc); D.Wdu = U.up(c.wDeltaU.a, rc);
    D.yref_s = U.up(c.yRef.a, rc); D.uref_s = U.up(c.uRef.a, rc);
    D.duref_s = U.up(c.duRef.a, rc); D.dmeas_s = U.up(c.dMeas.a, rc);
    {
        std::vector<double> lo0x(c.d.nx), hi0x(c.d.nx), lo0u(c.d.nu), hi0u(c.d.nu), lo0y(c.d.ny), hi0y(c.d.ny);
        for (int j = 0; j < c.d.nx; j++) { lo0x[j] = c.minX(j, 0); hi0x[j] = c.maxX(j, 0); }
        for (int j = 0; j < c.d.nu; j++) { lo0u[j] = c.minU(j, 0); hi0u[j] = c.maxU(j, 0); }
        for (int j = 0; j < c.d.ny; j++) { lo0y[j] = c.minY(j, 0); hi0y[j] = c.maxY(j, 0); }
        D.lo0x = U.up(lo0x, rc); D.hi0x = U.up(hi0x, rc); D.lo0u = U.up(lo0u, rc); D.hi0u = U.up(hi0u, rc);
        D.lo0y = U.up(lo0y, rc); D.hi0y = U.up(hi0y, rc);
        D.sX = U.up(c.sX, rc); D.sU = U.up(c.sU, rc);
    }
    if (o.big_n[0]) {          // structure-only condensing: room for what the device kernel computes
        D.H = U.reserve_big(o.big_n[0]); D.Kinv = U.reserve_big(o.big_n[1]); D.Gr = U.reserve_big(o.big_n[2]); D.Gc = U.reserve_big(o.big_n[3]); D.Y = U.reserve_big(o.big_n[4]);
    } else {
        D.H = U.up_big(o.H, rc); D.Kinv = U.up_big(o.Kinv, rc); D.Gr = U.up_big(o.Gr, rc); D.Gc = U.up_big(o.Gc, rc); D.Y = U.up_big(o.Y, rc);
    }
    D.lw = U.up(o.lw, rc); D.uw = U.up(o.uw, rc); D.rho_b = U.up(o.rho_b, rc);
    D.lg0 = U.up(o.lg0, rc); D.ug0 = U.up(o.ug0, rc); D.rho_g = U.up(o.rho_g, rc);
    D.g_kind = U.up(o.g_kind, rc); D.g_step = U.up(o.g_step, rc); D.g_comp = U.up(o.g_comp, rc); D.g_refrow = U.up(o.g_refrow, rc);
    {
        std::vector<int> fk, fs, fc; std::vector<double> fl, fh;
        for (auto &r : o.fixed_rows) { fk.push_back(r.kind); fs.push_back(r.step); fc.push_back(r.comp); fl.push_back(r.lo); fh.push_back(r.hi); }
        D.f_kind = U.up(fk, rc); D.f_step = U.up(fs, rc); D.f_comp = U.up(fc, rc); D.f_lo = U.up(fl, rc); D.f_hi = U.up(fh, rc);
    }
    D.boxrow_ptr = U.up(o.boxrow_ptr, rc); D.boxrow_ref = U.up(o.boxrow_ref, rc);
    D.boxrow_lo = U.up(o.boxrow_lo, rc); D.boxrow_hi = U.up(o.boxrow_hi, rc);
    D.blk = U.up(o.blk, rc);
    D.kin = o.kin; D.nxp = o.nxp; D.nup = o.nup; D.nyp = o.nyp; D.ione = o.ione;
    D.nz16 = o.nz16; D.mg16 = o.mg16; D.ns = o.ns; D.ns16 = o.ns16; D.kq16 = o.kq16; D.rowsA = o.rowsA; D.ldy16 = o.ldy16;
    D.fast_slice = mpcx::lmpc_fast_slice(D);
    D.MA0 = U.up(o.MA[0], rc); D.MA1 = U.up(o.MA[1], rc); D.Ym = U.up(o.Ym, rc);
    {
        std::vector<double> pk;
        auto packed = [&](const std::vector<double> &v, int rows, int K) -> const std::vector<double> & {
            pk.clear();
            if (!v.empty()) { pk.resize(mpcx::lmpc_packed_len(rows, K)); mpcx::lmpc_pack_mfma_tiles(v.data(), rows, K, pk.data()); }
            return pk;
        };
        D.MA0p = U.up(packed(o.MA[0], o.rowsA, o.kin), rc); D.MA1p = U.up(packed(o.MA[1], o.rowsA, o.kin), rc); D.Ymp = U.up(packed(o.Ym, o.ldy16, o.nz16), rc);
        std::vector<double> hpad;
        if (D.cost_direct && !o.MA[0].empty() && !o.H.empty()) {          // (a controller that takes lmpc_cost_mfma: the shared-model path with the cost from its definition)
            hpad.assign((size_t)o.nz16 * o.nz16, 0.0);
            for (int c = 0; c < o.nz; ++c)
                for (int r = 0; r < o.nz; ++r) hpad[(size_t)c * o.nz16 + r] = o.H[(size_t)c * o.ldz + r];
        }
        D.Hp = hpad.empty() ? nullptr : U.up(packed(hpad, o.nz16, o.nz16), rc);
    }
    D.MF0 = U.up(o.MF[0], rc); D.MF1 = U.up(o.MF[1], rc); D.rowsF = o.rowsF; D.nsp = o.nsp;
    // the fused kernel serves the one-chunk variant while the composed map stays small enough to stream from L2 per instance
    // (and the cost comes from the multipliers: otherwise the two-kernel path's batched cost kernel is the better one)
    D.fused_ok = (h->use_fused != 0 && mpcx::lmpc_kernel_variant(o.ldz, o.ldg) == 1 && o.rowsF <= 384 && !D.cost_direct) ? 1 : 0;
    // assemble + solve in one workgroup (lmpc_solve_group): the one-chunk variant, cost from the multipliers
    // lmpc_solve_group's LDS block: two box-bound vectors, sixteen per-instance slices, the staged operands, the result records -- one CU's worth at most
    // (otherwise the two-kernel path, which handled such a controller before the group form existed, takes it)
    // Round 5: the two-chunk variant as well (config 4's N = 50: eight instances per workgroup, the cost from its definition inside the solve);
    // there it is taken on request only (mpcx_lmpc_debug_use_fused(h, 2)): measured no faster than the two-kernel path
    const size_t ldsg = mpcx::lmpc_group_lds_bytes(D);
    const int cpv = mpcx::lmpc_kernel_variant(o.ldz, o.ldg);
    D.group_ok = (h->use_fused != 0 && ldsg > 0 && ldsg <= mpcx::lmpc_lds_limit() && ((cpv == 1 && !D.cost_direct) || (cpv == 2 && h->use_fused == 2))) ? 1 : 0;
    D.slo = U.up(o.slo, rc); D.shi = U.up(o.shi, rc);
}

// ---- heterogeneous batches: a bank of controllers solved together -------------------------------------------------------------
// In the reference every controller object owns its model (LMPC.hpp:751; ProblemBuilder.hpp:184-211 rebuilds the QP from it,
// :642-825).  A bank is K such objects -- same dimensions, same pattern of finite bounds, otherwise free: own A, B, C, weights,
// bounds, references, parameters -- condensed on the host cores in parallel and kept in one slab of HBM as K device structs; the
// kernels pick the struct of each instance (lmpc_model_of).  Per-model factors cannot be shared through L2: the assemble kernel
// reads its model's -Hinv and G Hinv once per instance, the solve the rows of Y in the working set -- this path is HBM-bound.
struct SlabUploader {
    static constexpr size_t kDevRegion = (size_t)1 << 44; // offsets from here on: the device-only region
    std::vector<char> host;                      // staging image of the uploaded region
    size_t dev_bytes = 0;                        // device-only region: arrays the condensing kernel fills (zeroed on the device, never staged)
    bool big_on_device = false;
    template <typename T>
    const T *up(const std::vector<T> &v, int &)
    {
        const size_t at = (host.size() + 15) / 16 * 16;
        host.resize(at + (v.size() ? v.size() : 1) * sizeof(T), 0);
        if (v.size()) std::memcpy(host.data() + at, v.data(), v.size() * sizeof(T));
        return reinterpret_cast<const T *>(at + 1);          // offset + 1 (so that offset 0 is not a null pointer); rebased below
    }
    const double *reserve_big(size_t n)
    {
        const size_t at = (dev_bytes + 15) / 16 * 16;
        dev_bytes = at + (n ? n : 1) * sizeof(double);
        return reinterpret_cast<const double *>(kDevRegion + at + 1);
    }
    template <typename T>
    const T *up_big(const std::vector<T> &v, int &rc)
    {
        if (!big_on_device) return up(v, rc);
        return reinterpret_cast<const T *>(reserve_big(v.size()));
    }
};

struct mpcx_lmpc_hetero {
    int device = 0, count = 0;
    mpcx_dims d{};
    mpcx::LmpcDev dev0{};                        // model 0 with device pointers (dimensions, LDS plan)
    mpcx::LmpcDev *models_d = nullptr;           // [count] device structs
    char *slab = nullptr, *slab_dev = nullptr;     // uploaded arrays / arrays the condensing kernel fills
    double *ws = nullptr; size_t ws_cap = 0;
    int active_words = 0, m_ref = 0;
    bool condensed_on_device = false;
    float setup_kernel_ms = 0, setup_total_ms = 0;   // the condensing kernel alone / the whole mpcx_lmpc_hetero_create
    double setup_flops = 0;                          // per controller (the host set-up's count)
    ~mpcx_lmpc_hetero()
    {
        if (models_d) (void)hipFree(models_d);
        if (slab) (void)hipFree(slab);
        if (slab_dev) (void)hipFree(slab_dev);
        if (ws) (void)hipFree(ws);
    }
};

static void rebase_dev(mpcx::LmpcDev &D, const char *base, const char *base_dev)
{
    // every pointer member was filled with (offset + 1) by SlabUploader: turn it into base + offset (of its region)
    auto fix = [&](auto &p) {
        using P = std::remove_reference_t<decltype(p)>;
        if (!p) return;
        const size_t off = reinterpret_cast<size_t>(p) - 1;
        p = off >= SlabUploader::kDevRegion ? reinterpret_cast<P>(base_dev + (off - SlabUploader::kDevRegion)) : reinterpret_cast<P>(base + off);
    };
    fix(D.A); fix(D.B); fix(D.C); fix(D.Bd); fix(D.Dd); fix(D.Wy); fix(D.Wu); fix(D.Wdu);
    fix(D.yref_s); fix(D.uref_s); fix(D.duref_s); fix(D.dmeas_s);
    fix(D.lo0x); fix(D.hi0x); fix(D.lo0u); fix(D.hi0u); fix(D.lo0y); fix(D.hi0y); fix(D.sX); fix(D.sU);
    fix(D.H); fix(D.Kinv); fix(D.Gr); fix(D.Gc); fix(D.Y); fix(D.lw); fix(D.uw); fix(D.rho_b); fix(D.lg0); fix(D.ug0); fix(D.rho_g);
    fix(D.g_kind); fix(D.g_step); fix(D.g_comp); fix(D.g_refrow);
    fix(D.f_kind); fix(D.f_step); fix(D.f_comp); fix(D.f_lo); fix(D.f_hi);
    fix(D.boxrow_ptr); fix(D.boxrow_ref); fix(D.boxrow_lo); fix(D.boxrow_hi); fix(D.blk);
    fix(D.MA0); fix(D.MA1); fix(D.Ym); fix(D.MA0p); fix(D.MA1p); fix(D.Ymp); fix(D.Hp); fix(D.slo); fix(D.shi); fix(D.MF0); fix(D.MF1);
}

extern "C" {

const char *mpcx_version(void) { return "mpcx 0.1.0 (gfx950)"; }
const char *mpcx_last_error(void) { return g_err.c_str(); }

void mpcx_lparams_default(mpcx_lparams *p)
{
    // mpc::LParameters defaults, reference include/mpc/Types.hpp:108-114,150-160
    p->maximum_iteration = 100; p->time_limit = 0; p->enable_warm_start = 0;
    p->alpha = 1.6; p->rho = 1e-6; p->eps_rel = 1e-4; p->eps_abs = 1e-4;
    p->eps_prim_inf = 1e-3; p->eps_dual_inf = 1e-3;
    p->verbose = 0; p->adaptive_rho = 1; p->polish = 1;
}

int mpcx_lmpc_create(const mpcx_dims *d, int device, mpcx_lmpc_t *out)
{
    if (!d || !out) return fail(MPCX_E_INVALID, "null argument");
    if (d->nx < 1 || d->nu < 1 || d->ny < 1 || d->ndu < 0 || d->ph < 1 || d->ch < 1 || d->ch > d->ph)
        return fail(MPCX_E_INVALID, "dimensions must satisfy nx,nu,ny,ph >= 1, 1 <= ch <= ph, ndu >= 0");
    if (d->nx > 64 || d->nu > 64 || d->ny > 64 || d->ndu > 64)
        return fail(MPCX_E_UNSUPPORTED, "nx, nu, ny, ndu must be <= 64 (one lane per component)");
    std::unique_ptr<mpcx_lmpc> h(new mpcx_lmpc(*d));
    h->device = device;
    h->host_only = device < 0;
    if (!h->host_only) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || device >= n)
            return fail(MPCX_E_DEVICE, "no such HIP device (the solve path has no CPU fallback)");
    }
    *out = h.release();
    return MPCX_OK;
}

int mpcx_lmpc_destroy(mpcx_lmpc_t h)
{
    if (!h) return MPCX_OK;
    if (!h->host_only) { (void)hipSetDevice(h->device); h->release(); h->release_staging(); }
    delete h;
    return MPCX_OK;
}

#define CHECK_H(h) do { if (!(h)) return fail(MPCX_E_INVALID, "null handle"); } while (0)

int mpcx_lmpc_set_state_space_model(mpcx_lmpc_t h, const double *A, const double *B, const double *C)
{
    CHECK_H(h);
    if (!A || !B || !C) return fail(MPCX_E_INVALID, "null matrix");
    auto &c = h->ctl;
    std::memcpy(c.A.a.data(), A, sizeof(double) * c.A.a.size());
    std::memcpy(c.B.a.data(), B, sizeof(double) * c.B.a.size());
    std::memcpy(c.C.a.data(), C, sizeof(double) * c.C.a.size());
    c.have_model = true;
    h->dirty = true;
    return MPCX_OK;
}

int mpcx_lmpc_set_disturbances(mpcx_lmpc_t h, const double *Bd, const double *Dd)
{
    CHECK_H(h);
    auto &c = h->ctl;
    if (c.d.ndu > 0) {
        if (!Bd || !Dd) return fail(MPCX_E_INVALID, "null matrix");
        std::memcpy(c.Bd.a.data(), Bd, sizeof(double) * c.Bd.a.size());
        std::memcpy(c.Dd.a.data(), Dd, sizeof(double) * c.Dd.a.size());
    }
    h->dirty = true;
    return MPCX_OK;
}

int mpcx_lmpc_set_objective_weights(mpcx_lmpc_t h, const double *OW, const double *UW, const double *DUW)
{
    CHECK_H(h);
    if (!OW || !UW || !DUW) return fail(MPCX_E_INVALID, "null matrix");
    h->ctl.set_objective(OW, UW, DUW);
    h->dirty = true;
    return MPCX_OK;
}

// Replicate-along-horizon helpers: {-1,-1} goes through the matrix setter, any other
// valid slice through the per-index setter, exactly as LMPC.hpp does (e.g. :436-481).
int mpcx_lmpc_set_objective_weights_slice(mpcx_lmpc_t h, const double *ow, const double *uw, const double *duw, int start, int end)
{
    CHECK_H(h);
    if (!ow || !uw || !duw) return fail(MPCX_E_INVALID, "null vector");
    auto &c = h->ctl;
    const int ph = c.d.ph;
    if (start == -1 && end == -1) {
        std::vector<double> O((size_t)c.d.ny * ph), U((size_t)c.d.nu * ph), D((size_t)c.d.nu * ph);
        for (int k = 0; k < ph; k++) {
            std::memcpy(&O[(size_t)k * c.d.ny], ow, sizeof(double) * c.d.ny);
            std::memcpy(&U[(size_t)k * c.d.nu], uw, sizeof(double) * c.d.nu);
            std::memcpy(&D[(size_t)k * c.d.nu], duw, sizeof(double) * c.d.nu);
        }
        c.set_objective(O.data(), U.data(), D.data());
    } else {
        if (!c.pred_slice_valid(start, end)) return fail(MPCX_E_INVALID, "The prediction horizon slice is out of bounds");
        for (int i = start; i < end; i++) c.set_objective_idx(i, ow, uw, duw);
    }
    h->dirty = true;
    return MPCX_OK;
}

int mpcx_lmpc_set_state_bounds(mpcx_lmpc_t h, const double *lo, const double *hi)
{
    CHECK_H(h);
    if (!lo || !hi) return fail(MPCX_E_INVALID, "null matrix");
    h->ctl.set_state_bounds(lo, hi);
    h->dirty = true;
    return MPCX_OK;
}
int mpcx_lmpc_set_state_bounds_slice(mpcx_lmpc_t h, const double *lo, const double *hi, int start, int end)
{
    CHECK_H(h);
    if (!lo || !hi) return fail(MPCX_E_INVALID, "null vector");
    auto &c = h->ctl;
    if (start == -1 && end == -1) {
        std::vector<double> L((size_t)c.d.nx * c.d.ph), H((size_t)c.d.nx * c.d.ph);
        for (int k = 0; k < c.d.ph; k++) {
            std::memcpy(&L[(size_t)k * c.d.nx], lo, sizeof(double) * c.d.nx);
            std::memcpy(&H[(size_t)k * c.d.nx], hi, sizeof(double) * c.d.nx);
        }
        c.set_state_bounds(L.data(), H.data());
    } else {
        if (!c.pred_slice_valid(start, end)) return fail(MPCX_E_INVALID, "The prediction horizon slice is out of bounds");
        for (int i = start; i < end; i++) c.set_state_bounds_idx(i, lo, hi);
    }
    h->dirty = true;
    return MPCX_OK;
}
int mpcx_lmpc_set_input_bounds(mpcx_lmpc_t h, const double *lo, const double *hi)
{
    CHECK_H(h);
    if (!lo || !hi) return fail(MPCX_E_INVALID, "null matrix");
    h->ctl.set_input_bounds(lo, hi);
    h->dirty = true;
    return MPCX_OK;
}
int mpcx_lmpc_set_input_bounds_slice(mpcx_lmpc_t h, const double *lo, const double *hi, int start, int end)
{
    CHECK_H(h);
    if (!lo || !hi) return fail(MPCX_E_INVALID, "null vector");
    auto &c = h->ctl;
    if (start == -1 && end == -1) {
        std::vector<double> L((size_t)c.d.nu * c.d.ch), H((size_t)c.d.nu * c.d.ch);
        for (int k = 0; k < c.d.ch; k++) {
            std::memcpy(&L[(size_t)k * c.d.nu], lo, sizeof(double) * c.d.nu);
            std::memcpy(&H[(size_t)k * c.d.nu], hi, sizeof(double) * c.d.nu);
        }
        c.set_input_bounds(L.data(), H.data());
    } else {
        if (!c.ctrl_slice_valid(start, end)) return fail(MPCX_E_INVALID, "The control horizon slice is out of bounds");
        for (int i = start; i < end; i++) c.set_input_bounds_idx(i, lo, hi);
    }
    h->dirty = true;
    return MPCX_OK;
}
int mpcx_lmpc_set_output_bounds(mpcx_lmpc_t h, const double *lo, const double *hi)
{
    CHECK_H(h);
    if (!lo || !hi) return fail(MPCX_E_INVALID, "null matrix");
    h->ctl.set_output_bounds(lo, hi);
    h->dirty = true;
    return MPCX_OK;
}
int mpcx_lmpc_set_output_bounds_slice(mpcx_lmpc_t h, const double *lo, const double *hi, int start, int end)
{
    CHECK_H(h);
    if (!lo || !hi) return fail(MPCX_E_INVALID, "null vector");
    auto &c = h->ctl;
    if (start == -1 && end == -1) {
        std::vector<double> L((size_t)c.d.ny * c.d.ph), H((size_t)c.d.ny * c.d.ph);
        for (int k = 0; k < c.d.ph; k++) {
            std::memcpy(&L[(size_t)k * c.d.ny], lo, sizeof(double) * c.d.ny);
            std::memcpy(&H[(size_t)k * c.d.ny], hi, sizeof(double) * c.d.ny);
        }
        c.set_output_bounds(L.data(), H.data());
    } else {
        if (!c.pred_slice_valid(start, end)) return fail(MPCX_E_INVALID, "The prediction horizon slice is out of bounds");
        for (int i = start; i < end; i++) c.set_output_bounds_idx(i, lo, hi);
    }
    h->dirty = true;
    return MPCX_OK;
}

int mpcx_lmpc_set_scalar_constraint_slice(mpcx_lmpc_t h, double smin, double smax, const double *X, const double *U, int start, int end)
{
    CHECK_H(h);
    if (!X || !U) return fail(MPCX_E_INVALID, "null vector");
    auto &c = h->ctl;
    if (start == -1 && end == -1) {
        std::vector<double> L(c.d.ph, smin), H(c.d.ph, smax);
        c.set_scalar_vec(L.data(), H.data(), X, U);
    } else {
        if (!c.pred_slice_valid(start, end)) return fail(MPCX_E_INVALID, "The prediction horizon slice is out of bounds");
        for (int i = start; i < end; i++) c.set_scalar_idx(i, smin, smax, X, U);
    }
    h->dirty = true;
    return MPCX_OK;
}
int mpcx_lmpc_set_scalar_constraint_index(mpcx_lmpc_t h, int index, double smin, double smax, const double *X, const double *U)
{
    CHECK_H(h);
    if (!X || !U) return fail(MPCX_E_INVALID, "null vector");
    if (index < 0 || index >= h->ctl.d.ph) return fail(MPCX_E_INVALID, "Horizon index out of bounds");
    h->ctl.set_scalar_idx(index, smin, smax, X, U);
    h->dirty = true;
    return MPCX_OK;
}

int mpcx_lmpc_set_references(mpcx_lmpc_t h, const double *yref, const double *uref, const double *duref)
{
    CHECK_H(h);
    if (!yref || !uref || !duref) return fail(MPCX_E_INVALID, "null matrix");
    auto &c = h->ctl;
    std::memcpy(c.yRef.a.data(), yref, sizeof(double) * c.yRef.a.size());
    std::memcpy(c.uRef.a.data(), uref, sizeof(double) * c.uRef.a.size());
    std::memcpy(c.duRef.a.data(), duref, sizeof(double) * c.duRef.a.size());
    h->refs_dirty = true;        // the time-invariant terms stay (the reference's setReferences does not rebuild them either)
    return MPCX_OK;
}
int mpcx_lmpc_set_references_slice(mpcx_lmpc_t h, const double *yref, const double *uref, const double *duref, int start, int end)
{
    CHECK_H(h);
    if (!yref || !uref || !duref) return fail(MPCX_E_INVALID, "null vector");
    auto &c = h->ctl;
    int s = start, e = end;
    if (start == -1 && end == -1) { s = 0; e = c.d.ph; }
    else if (!c.pred_slice_valid(start, end)) return fail(MPCX_E_INVALID, "The prediction horizon slice is out of bounds");
    for (int i = s; i < e; i++) {
        std::memcpy(c.yRef.col(i), yref, sizeof(double) * c.d.ny);
        std::memcpy(c.uRef.col(i), uref, sizeof(double) * c.d.nu);
        std::memcpy(c.duRef.col(i), duref, sizeof(double) * c.d.nu);
    }
    h->refs_dirty = true;        // the time-invariant terms stay (the reference's setReferences does not rebuild them either)
    return MPCX_OK;
}
int mpcx_lmpc_set_exogenous_inputs(mpcx_lmpc_t h, const double *dmeas)
{
    CHECK_H(h);
    auto &c = h->ctl;
    if (c.d.ndu > 0) {
        if (!dmeas) return fail(MPCX_E_INVALID, "null matrix");
        std::memcpy(c.dMeas.a.data(), dmeas, sizeof(double) * c.dMeas.a.size());
    }
    h->refs_dirty = true;        // the time-invariant terms stay (the reference's setReferences does not rebuild them either)
    return MPCX_OK;
}
int mpcx_lmpc_set_exogenous_inputs_slice(mpcx_lmpc_t h, const double *dmeas, int start, int end)
{
    CHECK_H(h);
    auto &c = h->ctl;
    int s = start, e = end;
    if (start == -1 && end == -1) { s = 0; e = c.d.ph; }
    // the reference validates this slice against the CONTROL horizon (LMPC.hpp:571, isControlHorizonSliceValid)
    else if (!c.ctrl_slice_valid(start, end)) return fail(MPCX_E_INVALID, "The control horizon slice is out of bounds");
    if (c.d.ndu > 0) {
        if (!dmeas) return fail(MPCX_E_INVALID, "null vector");
        for (int i = s; i < e; i++) std::memcpy(c.dMeas.col(i), dmeas, sizeof(double) * c.d.ndu);
    }
    h->refs_dirty = true;        // the time-invariant terms stay (the reference's setReferences does not rebuild them either)
    return MPCX_OK;
}

int mpcx_lmpc_set_optimizer_parameters(mpcx_lmpc_t h, const mpcx_lparams *p)
{
    CHECK_H(h);
    if (!p) return fail(MPCX_E_INVALID, "null parameters");
    if (p->maximum_iteration < 0 || !(p->alpha > 0 && p->alpha < 2) || !(p->rho > 0))
        return fail(MPCX_E_INVALID, "parameters out of range");
    h->ctl.prm = *p;
    h->dirty = true;
    return MPCX_OK;
}

int mpcx_lmpc_set_strict_infeasibility(mpcx_lmpc_t h, int on)
{
    CHECK_H(h);
    h->strict_infeasible = on != 0;
    h->dirty = true;
    return MPCX_OK;
}

// References / exogenous inputs changed and nothing else: the condensed matrices, their factors, the workspace and the
// queues stay where they are; what depends on the references -- the shared reference arrays and the constant column of the
// stacked maps of the MFMA assemble kernel -- is recomputed on the host (one roll-out per input component) and copied over
// the device arrays in place.
static int refresh_references(mpcx_lmpc_t h)
{
    const auto &c = h->ctl;
    h->ctl.refresh_fast_maps(h->cond);
    h->refs_dirty = false;
    ++h->n_ref_refreshes;
    if (h->host_only) return MPCX_OK;
    if (hipSetDevice(h->device) != hipSuccess) return fail(MPCX_E_DEVICE, "hipSetDevice failed");
    // the arrays are overwritten in place: launches of this handle still in flight on a non-blocking stream (they read them) are not
    // ordered against a copy on the null stream -- wait for the device first (a reference change is a host-side event, not the hot path)
    if (hipDeviceSynchronize() != hipSuccess) return fail(MPCX_E_DEVICE, "hipDeviceSynchronize failed");
    const mpcx::LmpcDev &D = h->dev;
    const bool ok = h->reup(D.yref_s, c.yRef.a) && h->reup(D.uref_s, c.uRef.a) && h->reup(D.duref_s, c.duRef.a) &&
                    h->reup(D.dmeas_s, c.dMeas.a) && h->reup(D.MA0, h->cond.MA[0]) && h->reup(D.MA1, h->cond.MA[1]) &&
                    h->reup(D.MF0, h->cond.MF[0]) && h->reup(D.MF1, h->cond.MF[1]);
    if (ok && !h->cond.MA[0].empty()) {          // ... and their packed copies (lmpc_solve_group's)
        std::vector<double> pk(mpcx::lmpc_packed_len(h->cond.rowsA, h->cond.kin));
        for (int v = 0; v < 2; ++v) {
            mpcx::lmpc_pack_mfma_tiles(h->cond.MA[v].data(), h->cond.rowsA, h->cond.kin, pk.data());
            if (!h->reup(v ? D.MA1p : D.MA0p, pk)) return fail(MPCX_E_DEVICE, "device upload failed");
        }
    }
    return ok ? MPCX_OK : fail(MPCX_E_DEVICE, "device upload failed");
}

int mpcx_lmpc_setup(mpcx_lmpc_t h)
{
    CHECK_H(h);
    if (!h->dirty) return h->refs_dirty ? refresh_references(h) : MPCX_OK;
    h->refs_dirty = false;
    ++h->n_full_setups;
    std::string msg = h->ctl.condense(h->cond);
    if (!msg.empty()) return fail(msg == "state-space model not set" ? MPCX_E_STATE : MPCX_E_NUMERIC, msg);
    const auto &c = h->ctl;
    const auto &o = h->cond;
    if (mpcx::lmpc_kernel_variant(o.ldz, o.ldg) < 0)
        return fail(MPCX_E_UNSUPPORTED, "condensed problem larger than 512 variables / rows");
    mpcx::LmpcDev &D = h->dev;
    fill_dev_scalars(h, c, o, D);
    if (h->host_only) { h->dirty = false; return MPCX_OK; }

    if (hipSetDevice(h->device) != hipSuccess) return fail(MPCX_E_DEVICE, "hipSetDevice failed");
    h->release();
    int rc = MPCX_OK;
    fill_dev_arrays(h, c, o, D, *h, rc);
    if (rc != MPCX_OK) return fail(rc, "device upload failed");
    {
        void *p = nullptr;
        if (hipMalloc(&p, sizeof(mpcx::LmpcDev)) != hipSuccess) return fail(MPCX_E_DEVICE, "hipMalloc failed");
        h->allocs.push_back(p);
        if (hipMemcpy(p, &D, sizeof(mpcx::LmpcDev), hipMemcpyHostToDevice) != hipSuccess) return fail(MPCX_E_DEVICE, "hipMemcpy failed");
        h->dev_d = static_cast<mpcx::LmpcDev *>(p);
    }
    h->dirty = false;
    return MPCX_OK;
}

static int make_batch(mpcx_lmpc_t h, const mpcx_lmpc_batch *b, mpcx::LmpcBatchDev &B)
{
    const auto &d = h->ctl.d;
    const auto &D = h->dev;
    if (b->batch < 0) return fail(MPCX_E_INVALID, "negative batch");
    if (b->batch > 0 && (!b->x0 || !b->u0 || !b->cmd)) return fail(MPCX_E_INVALID, "x0, u0 and cmd are required");
    B = mpcx::LmpcBatchDev{};
    B.batch = b->batch; B.x0 = b->x0; B.u0 = b->u0;
    auto refsel = [&](const double *p, int mode, const double *shared, int n, const double *&op, long &bs, long &ks) -> bool {
        if (mode == MPCX_REF_SHARED) { op = shared; bs = 0; ks = n; return true; }
        if (!p) return false;
        if (mode == MPCX_REF_PER_INSTANCE) { op = p; bs = n; ks = 0; return true; }
        if (mode == MPCX_REF_PER_STEP) { op = p; bs = (long)d.ph * n; ks = n; return true; }
        return false;
    };
    if (!refsel(b->yref, b->yref_mode, D.yref_s, d.ny, B.yref, B.yref_bs, B.yref_ks) ||
        !refsel(b->uref, b->uref_mode, D.uref_s, d.nu, B.uref, B.uref_bs, B.uref_ks) ||
        !refsel(b->duref, b->duref_mode, D.duref_s, d.nu, B.duref, B.duref_bs, B.duref_ks) ||
        !refsel(b->dmeas, b->dmeas_mode, D.dmeas_s, d.ndu, B.dmeas, B.dmeas_bs, B.dmeas_ks))
        return fail(MPCX_E_INVALID, "reference array missing for a non-shared mode, or unknown mode");
    B.cmd = b->cmd; B.cost = b->cost; B.status = b->status; B.solver_status = b->solver_status;
    B.is_feasible = b->is_feasible; B.iterations = b->iterations;
    B.active_lower = b->active_lower; B.active_upper = b->active_upper;
    B.seq_state = b->seq_state; B.seq_output = b->seq_output; B.seq_input = b->seq_input;
    B.polish_rounds = b->polish_rounds; B.active_count = b->active_count;
    B.warm_lower = b->warm_active_lower; B.warm_upper = b->warm_active_upper; B.warm_shift = b->warm_shift;
    if ((B.warm_lower == nullptr) != (B.warm_upper == nullptr)) return fail(MPCX_E_INVALID, "warm_active_lower and warm_active_upper go together");
    B.dbg_cycles = h->dbg_cycles;
    return MPCX_OK;
}

int mpcx_lmpc_solve_batch(mpcx_lmpc_t h, const mpcx_lmpc_batch *b, void *stream)
{
    CHECK_H(h);
    if (!b) return fail(MPCX_E_INVALID, "null batch");
    if (h->host_only) return fail(MPCX_E_DEVICE, "host-only handle: the solve path needs a HIP device, there is no CPU fallback");
    int rc = mpcx_lmpc_setup(h);
    if (rc != MPCX_OK) return rc;
    if (hipSetDevice(h->device) != hipSuccess) return fail(MPCX_E_DEVICE, "hipSetDevice failed");
    if (b->batch == 0) return MPCX_OK;      // an empty batch is a no-op, whatever its pointers
    mpcx::LmpcBatchDev B;
    rc = make_batch(h, b, B);
    if (rc != MPCX_OK) return rc;
    if ((size_t)b->batch > h->ws_cap) {
        // grows only when a larger batch than ever before arrives (not capturable in a graph)
        if (h->ws) (void)hipFree(h->ws);
        if (h->done) (void)hipFree(h->done);
        h->ws = nullptr; h->done = nullptr; h->ws_cap = 0;
        if (hipMalloc(reinterpret_cast<void **>(&h->ws), (size_t)b->batch * h->dev.wsld * sizeof(double)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&h->done), (size_t)b->batch * sizeof(int32_t)) != hipSuccess)
            return fail(MPCX_E_DEVICE, "workspace allocation failed");
        if (!h->pcounter) {
            if (hipMalloc(reinterpret_cast<void **>(&h->pcounter), 8 * sizeof(int)) != hipSuccess) return fail(MPCX_E_DEVICE, "workspace allocation failed");
            (void)hipMemset(h->pcounter, 0, 8 * sizeof(int));
        }
        h->ws_cap = (size_t)b->batch;
    }
    // the MFMA assemble kernel serves shared or per-instance-constant output references with
    // everything else shared; any other layout goes through the generic roll-out kernel
    int fast = -1;
    if (b->uref_mode == MPCX_REF_SHARED && b->duref_mode == MPCX_REF_SHARED && b->dmeas_mode == MPCX_REF_SHARED && !h->force_generic) {
        if (b->yref_mode == MPCX_REF_SHARED) fast = 0;
        else if (b->yref_mode == MPCX_REF_PER_INSTANCE) fast = 1;
    }
    if (fast >= 0 && h->dev.group_ok && (h->use_fused == 2 || (h->use_fused < 0 && (b->batch > h->total_batch ? b->batch : h->total_batch) <= h->group_max))) { B.fused = fast + 3; B.done = h->done; }
    else if (fast >= 0 && h->dev.fused_ok && !B.dbg_cycles && h->use_fused == 1) { B.fused = fast + 1; B.pcounter = h->pcounter; }
    int lr = mpcx::lmpc_launch(h->dev, h->dev_d, B, h->ws, stream, 7, fast);
    if (lr == -2) return fail(MPCX_E_UNSUPPORTED, "problem dimensions exceed the kernel's LDS budget");
    if (lr != 0) return fail(MPCX_E_DEVICE, std::string("kernel launch failed: ") + hipGetErrorString(hipGetLastError()));
    return MPCX_OK;
}

struct mpcx_lmpc_graph {
    hipGraphExec_t exec = nullptr;
    int device = 0;
    mpcx_lmpc_t owner = nullptr;        // the controller whose step was captured
};

int mpcx_lmpc_graph_create(mpcx_lmpc_t h, const mpcx_lmpc_batch *b, void *stream, mpcx_lmpc_graph_t *out)
{
    CHECK_H(h);
    if (!b || !out) return fail(MPCX_E_INVALID, "null argument");
    if (!stream) return fail(MPCX_E_INVALID, "a graph is captured on a non-default stream");
    if (b->batch <= 0) return fail(MPCX_E_INVALID, "empty batch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // one plain solve first: it runs whatever set-up is pending, sizes the workspace and configures the kernels -- none of
    // which can be part of a capture
    int rc = mpcx_lmpc_solve_batch(h, b, stream);
    if (rc != MPCX_OK) return rc;
    if (hipStreamSynchronize(s) != hipSuccess) return fail(MPCX_E_DEVICE, "the warm-up solve failed");
    if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess) return fail(MPCX_E_DEVICE, "hipStreamBeginCapture failed");
    rc = mpcx_lmpc_solve_batch(h, b, stream);
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(s, &graph);
    if (rc != MPCX_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (ec != hipSuccess || !graph) return fail(MPCX_E_DEVICE, "hipStreamEndCapture failed");
    auto *g = new mpcx_lmpc_graph;
    g->device = h->device;
    g->owner = h;
    const hipError_t ei = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) { delete g; return fail(MPCX_E_DEVICE, "hipGraphInstantiate failed"); }
    *out = g;
    return MPCX_OK;
}

int mpcx_lmpc_graph_launch(mpcx_lmpc_graph_t g, void *stream)
{
    if (!g || !g->exec) return fail(MPCX_E_INVALID, "null graph");
    // a setter since the capture needs a set-up pass the captured launches do not contain (mpcx_lmpc_solve_batch runs it; a
    // reference-only change is picked up by one plain solve, anything else needs a new graph)
    if (g->owner && (g->owner->dirty || g->owner->refs_dirty))
        return fail(MPCX_E_STATE, "the controller changed since the graph was captured: call mpcx_lmpc_solve_batch once (references) or capture again");
    if (hipSetDevice(g->device) != hipSuccess) return fail(MPCX_E_DEVICE, "hipSetDevice failed");
    if (hipGraphLaunch(g->exec, reinterpret_cast<hipStream_t>(stream)) != hipSuccess) return fail(MPCX_E_DEVICE, "hipGraphLaunch failed");
    return MPCX_OK;
}

int mpcx_lmpc_graph_destroy(mpcx_lmpc_graph_t g)
{
    if (!g) return MPCX_OK;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    delete g;
    return MPCX_OK;
}

int mpcx_lmpc_solve_host(mpcx_lmpc_t h, int batch, const double *x0, const double *u0,
                         double *cmd, double *cost, int32_t *status, int32_t *solver_status, int32_t *is_feasible,
                         double *seq_state, double *seq_output, double *seq_input)
{
    CHECK_H(h);
    if (batch < 0 || (batch > 0 && (!x0 || !u0 || !cmd))) return fail(MPCX_E_INVALID, "x0, u0 and cmd are required");
    if (h->host_only) return fail(MPCX_E_DEVICE, "host-only handle: the solve path needs a HIP device, there is no CPU fallback");
    if (batch == 0) return MPCX_OK;
    int rc = mpcx_lmpc_setup(h);                 // active_words below comes from the condensed model
    if (rc != MPCX_OK) return rc;
    if (hipSetDevice(h->device) != hipSuccess) return fail(MPCX_E_DEVICE, "hipSetDevice failed");
    const auto &d = h->ctl.d;
    const size_t B = (size_t)batch, n1 = (size_t)d.ph + 1, aw = (size_t)h->dev.active_words;
    const size_t per = d.nx + d.nu + d.nu + 1 + n1 * (d.nx + d.ny + d.nu);
    if (B > h->stage_cap) {                       // staging buffers live in the handle: no allocation in the steady state
        h->release_staging();
        if (hipMalloc(reinterpret_cast<void **>(&h->stage_d), B * per * sizeof(double)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&h->stage_i), B * 4 * sizeof(int32_t)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&h->stage_act), B * 4 * aw * sizeof(uint32_t)) != hipSuccess) {
            h->release_staging();
            return fail(MPCX_E_DEVICE, "staging allocation failed");
        }
        h->stage_cap = B;
    }
    double *dx0 = h->stage_d, *du0 = dx0 + B * d.nx, *dcmd = du0 + B * d.nu, *dcost = dcmd + B * d.nu;
    double *dss = dcost + B, *dso = dss + B * n1 * d.nx, *dsi = dso + B * n1 * d.ny;
    int32_t *ibuf = h->stage_i;
    if (hipMemcpy(dx0, x0, B * d.nx * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(du0, u0, B * d.nu * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        return fail(MPCX_E_DEVICE, "copy of x0 / u0 to the device failed");
    mpcx_lmpc_batch b{};
    b.batch = batch; b.x0 = dx0; b.u0 = du0;
    b.cmd = dcmd; b.cost = dcost;
    b.status = ibuf; b.solver_status = ibuf + B; b.is_feasible = ibuf + 2 * B; b.iterations = ibuf + 3 * B;
    const bool want_seq = seq_state || seq_output || seq_input;
    if (want_seq) { b.seq_state = dss; b.seq_output = dso; b.seq_input = dsi; }
    // LParameters::enable_warm_start (LOptimizer.hpp:268-281, LMPC.hpp:677-722): the reference re-uses the previous call's
    // primal / dual pair; what carries over here is the previous call's active set, shifted one step (receding horizon).
    // Two pairs of bitmaps alternate between "previous" and "current".
    if (h->ctl.prm.enable_warm_start) {
        uint32_t *cur = h->stage_act, *prev = h->stage_act + 2 * B * aw;
        if (h->warm_batch == batch) {
            if (hipMemcpy(prev, cur, 2 * B * aw * sizeof(uint32_t), hipMemcpyDeviceToDevice) != hipSuccess)
                return fail(MPCX_E_DEVICE, "device copy failed");
            b.warm_active_lower = prev; b.warm_active_upper = prev + B * aw; b.warm_shift = 1;
        }
        b.active_lower = cur; b.active_upper = cur + B * aw;
    }
    rc = mpcx_lmpc_solve_batch(h, &b, nullptr);
    if (rc == MPCX_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(MPCX_E_DEVICE, "kernel execution failed");
    h->warm_batch = (rc == MPCX_OK && h->ctl.prm.enable_warm_start) ? batch : 0;
    if (rc == MPCX_OK) {
        bool ok = true;
        auto back = [&](void *dst, const void *src, size_t bytes) { if (dst) ok = ok && hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess; };
        back(cmd, dcmd, B * d.nu * sizeof(double)); back(cost, dcost, B * sizeof(double));
        back(status, ibuf, B * sizeof(int32_t)); back(solver_status, ibuf + B, B * sizeof(int32_t)); back(is_feasible, ibuf + 2 * B, B * sizeof(int32_t));
        back(seq_state, dss, B * n1 * d.nx * sizeof(double)); back(seq_output, dso, B * n1 * d.ny * sizeof(double));
        back(seq_input, dsi, B * n1 * d.nu * sizeof(double));
        if (!ok) rc = fail(MPCX_E_DEVICE, "copy of the results to the host failed");
    }
    return rc;
}

int mpcx_lmpc_time_solve_batch(mpcx_lmpc_t h, const mpcx_lmpc_batch *b, void *stream, int repeats, float *ms_mean)
{
    CHECK_H(h);
    if (!b || !ms_mean || repeats < 1) return fail(MPCX_E_INVALID, "bad argument");
    if (h->host_only) return fail(MPCX_E_DEVICE, "host-only handle");
    int rc = mpcx_lmpc_setup(h);
    if (rc != MPCX_OK) return rc;
    if (hipSetDevice(h->device) != hipSuccess) return fail(MPCX_E_DEVICE, "hipSetDevice failed");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(MPCX_E_DEVICE, "hipEventCreate failed");
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < repeats; i++) {
        rc = mpcx_lmpc_solve_batch(h, b, stream);
        if (rc != MPCX_OK) break;
    }
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_mean = ms / (float)repeats;
    return rc;
}

int mpcx_lmpc_get_info(mpcx_lmpc_t h, mpcx_lmpc_info *info)
{
    CHECK_H(h);
    if (!info) return fail(MPCX_E_INVALID, "null info");
    int rc = mpcx_lmpc_setup(h);
    if (rc != MPCX_OK) return rc;
    const auto &o = h->cond;
    const auto &d = h->ctl.d;
    info->n_ref = o.n_ref; info->m_ref = o.m_ref; info->neq_ref = o.neq_ref;
    info->nz = o.nz; info->mg = o.mg; info->active_words = o.active_words;
    info->kernel_variant = mpcx::lmpc_kernel_variant(o.ldz, o.ldg);
    info->flops_setup = o.flops_setup;
    const double nz = o.nz, mg = o.mg;
    info->flops_per_admm_iter = 2 * nz * nz + 4 * mg * nz + 12 * nz + 10 * mg;
    info->flops_fixed_per_solve = 2.0 * d.ph * d.nx * d.nx + 2.0 * (d.ph + 1) * d.ny * d.nx +
                                  2.0 * d.ph * (d.nx * d.nx + d.ny * d.nx + d.nx * d.nu) +
                                  2.0 * nz * (nz + mg) + 2.0 * nz * nz;
    info->bytes_per_solve = 8.0 * (d.nx + d.nu) + 8.0 * d.nu + 8.0 + 16.0;
    return MPCX_OK;
}

/* ---- heterogeneous batches -------------------------------------------------------------------------------------------------- */
int mpcx_lmpc_hetero_create(const mpcx_lmpc_t *controllers, int count, int device, mpcx_lmpc_hetero_t *out)
{
    return mpcx_lmpc_hetero_create_ex(controllers, count, device, 0, out);
}

int mpcx_lmpc_hetero_create_ex(const mpcx_lmpc_t *controllers, int count, int device, int condense_on_host, mpcx_lmpc_hetero_t *out)
{
    if (!controllers || !out || count < 1) return fail(MPCX_E_INVALID, "need at least one controller");
    const auto t_begin = std::chrono::steady_clock::now();
    for (int k = 0; k < count; ++k) if (!controllers[k]) return fail(MPCX_E_INVALID, "null controller in the bank");
    int ndev = 0;
    if (device < 0 || hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev)
        return fail(MPCX_E_DEVICE, "no such HIP device (the solve path has no CPU fallback)");
    // Controller 0 is condensed in full on the host: it fixes the structure (dimensions, constraint rows) and says whether the
    // dimensions fit the device kernel.  The others: in full too (condense_on_host, or dimensions beyond the kernel's LDS plan), or
    // their structure and O(n) arrays only -- the O(n^3) arrays of all K are then computed by lmpc_condense_models.
    std::vector<mpcx::Condensed> cond((size_t)count);
    std::vector<std::string> errs((size_t)count);
    errs[0] = controllers[0]->ctl.condense(cond[0]);
    if (!errs[0].empty()) return fail(MPCX_E_NUMERIC, "controller 0: " + errs[0]);
    bool on_device = !condense_on_host;
    {
        mpcx::LmpcDev probe{};
        probe.nz = cond[0].nz; probe.mg = cond[0].mg; probe.nx = controllers[0]->ctl.d.nx; probe.ny = controllers[0]->ctl.d.ny;
        if (mpcx::lmpc_condense_lds(probe, nullptr, nullptr) > mpcx::lmpc_lds_limit() - 64) on_device = false;
    }
    const mpcx::Condensed *like = on_device ? &cond[0] : nullptr;
    {
        unsigned nt = std::thread::hardware_concurrency();
        if (nt < 1) nt = 1;
        if (nt > 32) nt = 32;
        if ((int)nt > count) nt = (unsigned)count;
        std::atomic<int> next{0};
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t)
            pool.emplace_back([&]() {
                for (int k = next.fetch_add(1) + 1; k < count; k = next.fetch_add(1) + 1) errs[(size_t)k] = controllers[k]->ctl.condense(cond[(size_t)k], like);
            });
        for (auto &th : pool) th.join();
    }
    for (int k = 0; k < count; ++k)
        if (!errs[(size_t)k].empty()) return fail(MPCX_E_NUMERIC, "controller " + std::to_string(k) + ": " + errs[(size_t)k]);
    // one structure for all: dimensions, constraint rows (which bounds are finite), move blocking
    const auto &c0 = controllers[0]->ctl;
    const auto &o0 = cond[0];
    if (mpcx::lmpc_kernel_variant(o0.ldz, o0.ldg) < 0) return fail(MPCX_E_UNSUPPORTED, "condensed problem larger than 512 variables / rows");
    for (int k = 1; k < count; ++k) {
        const auto &ck = controllers[k]->ctl;
        const auto &ok = cond[(size_t)k];
        const bool same = ck.d.nx == c0.d.nx && ck.d.nu == c0.d.nu && ck.d.ny == c0.d.ny && ck.d.ndu == c0.d.ndu && ck.d.ph == c0.d.ph && ck.d.ch == c0.d.ch &&
                          ok.nz == o0.nz && ok.mg == o0.mg && ok.g_refrow == o0.g_refrow && ok.boxrow_ptr == o0.boxrow_ptr && ok.boxrow_ref == o0.boxrow_ref &&
                          ok.fixed_rows.size() == o0.fixed_rows.size() && ok.blk == o0.blk;
        if (!same) return fail(MPCX_E_INVALID, "controller " + std::to_string(k) + " differs from controller 0 in dimensions or in which bounds are finite");
    }
    std::unique_ptr<mpcx_lmpc_hetero> f(new mpcx_lmpc_hetero);
    f->device = device; f->count = count; f->d = c0.d; f->active_words = o0.active_words; f->m_ref = o0.m_ref;
    if (hipSetDevice(device) != hipSuccess) return fail(MPCX_E_DEVICE, "hipSetDevice failed");
    SlabUploader U;
    U.big_on_device = on_device;
    std::vector<mpcx::LmpcDev> devs((size_t)count);
    int rc = MPCX_OK;
    for (int k = 0; k < count; ++k) {
        auto &o = cond[(size_t)k];
        // the stacked maps of the MFMA assemble kernel serve sixteen instances of ONE model: not used here, not uploaded
        o.MA[0].clear(); o.MA[1].clear(); o.MF[0].clear(); o.MF[1].clear(); o.Ym.clear();
        fill_dev_scalars(controllers[k], controllers[k]->ctl, o, devs[(size_t)k]);
        fill_dev_arrays(controllers[k], controllers[k]->ctl, o, devs[(size_t)k], U, rc);
        devs[(size_t)k].fused_ok = 0; devs[(size_t)k].group_ok = 0;
    }
    if (hipMalloc(reinterpret_cast<void **>(&f->slab), U.host.size()) != hipSuccess ||
        hipMemcpy(f->slab, U.host.data(), U.host.size(), hipMemcpyHostToDevice) != hipSuccess)
        return fail(MPCX_E_DEVICE, "could not upload the bank (" + std::to_string(U.host.size() >> 20) + " MiB)");
    if (U.dev_bytes) {
        if (hipMalloc(reinterpret_cast<void **>(&f->slab_dev), U.dev_bytes) != hipSuccess || hipMemset(f->slab_dev, 0, U.dev_bytes) != hipSuccess)
            return fail(MPCX_E_DEVICE, "could not allocate the bank's factors (" + std::to_string(U.dev_bytes >> 20) + " MiB)");
    }
    for (auto &D : devs) rebase_dev(D, f->slab, f->slab_dev);
    if (hipMalloc(reinterpret_cast<void **>(&f->models_d), sizeof(mpcx::LmpcDev) * (size_t)count) != hipSuccess ||
        hipMemcpy(f->models_d, devs.data(), sizeof(mpcx::LmpcDev) * (size_t)count, hipMemcpyHostToDevice) != hipSuccess)
        return fail(MPCX_E_DEVICE, "could not upload the bank's model table");
    f->dev0 = devs[0];
    f->setup_flops = cond[0].flops_setup;
    if (on_device) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, nullptr);
        const int lr = mpcx::lmpc_condense_launch(f->models_d, f->dev0, count, nullptr);
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&f->setup_kernel_ms, e0, e1);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        if (lr != 0) return fail(MPCX_E_DEVICE, "the device condensing kernel could not be launched (" + std::to_string(lr) + ")");
        const hipError_t es = hipDeviceSynchronize();
        if (es != hipSuccess) return fail(MPCX_E_DEVICE, std::string("the device condensing kernel failed: ") + hipGetErrorString(es));
        // what LmpcController::condense reports on the host (MPCX_E_NUMERIC for a Hessian / ADMM matrix that does not factor), and what the
        // host path reports as "differs from controller 0" (a constraint row that vanishes in one controller but not in the first)
        if (hipMemcpy(devs.data(), f->models_d, sizeof(mpcx::LmpcDev) * (size_t)count, hipMemcpyDeviceToHost) != hipSuccess)
            return fail(MPCX_E_DEVICE, "could not read the bank's model table back");
        for (int k = 0; k < count; ++k) {
            const int cs = devs[(size_t)k].cond_status;
            if (cs & 1) return fail(MPCX_E_NUMERIC, "controller " + std::to_string(k) + ": condensed Hessian is not positive semidefinite");
            if (cs & 2) return fail(MPCX_E_NUMERIC, "controller " + std::to_string(k) + ": ADMM matrix is not positive definite");
            if (cs & 4) return fail(MPCX_E_INVALID, "controller " + std::to_string(k) + " differs from controller 0: one of its constraint rows does not depend on the inputs");
        }
    }
    f->condensed_on_device = on_device;
    f->setup_total_ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    *out = f.release();
    return MPCX_OK;
}

int mpcx_lmpc_hetero_destroy(mpcx_lmpc_hetero_t f)
{
    if (!f) return MPCX_OK;
    (void)hipSetDevice(f->device);
    delete f;
    return MPCX_OK;
}

/* testing aid: one O(n^3) array ("H", "Kinv", "Gr", "Gc", "Y", "rho_b", "rho_g") of model k copied to the host; returns its length */
int mpcx_lmpc_hetero_debug_get(mpcx_lmpc_hetero_t f, int k, const char *name, double *out, int cap)
{
    if (!f || k < 0 || k >= f->count || !name) return fail(MPCX_E_INVALID, "bad argument");
    if (hipSetDevice(f->device) != hipSuccess) return fail(MPCX_E_DEVICE, "hipSetDevice failed");
    mpcx::LmpcDev D;
    if (hipMemcpy(&D, f->models_d + k, sizeof(D), hipMemcpyDeviceToHost) != hipSuccess) return fail(MPCX_E_DEVICE, "hipMemcpy failed");
    const std::string n(name);
    const double *src = nullptr; size_t len = 0;
    if (n == "H") { src = D.H; len = (size_t)D.ldz * D.ldz; } else if (n == "Kinv") { src = D.Kinv; len = (size_t)D.ldz * D.ldz; }
    else if (n == "Gr") { src = D.Gr; len = (size_t)D.ldg * D.ldz; } else if (n == "Gc") { src = D.Gc; len = (size_t)D.ldz * D.ldg; }
    else if (n == "Y") { src = D.Y; len = (size_t)D.ldy * D.ldy; } else if (n == "rho_b") { src = D.rho_b; len = (size_t)D.ldz; }
    else if (n == "rho_g") { src = D.rho_g; len = (size_t)D.ldg; }
    else if (n == "flags") {       // cost_direct, condensed on the device, condensing kernel ms, whole create ms, set-up flops per controller
        if (out && cap >= 5) { out[0] = D.cost_direct; out[1] = f->condensed_on_device ? 1.0 : 0.0; out[2] = f->setup_kernel_ms; out[3] = f->setup_total_ms; out[4] = f->setup_flops; }
        return 5;
    }
    else return fail(MPCX_E_INVALID, "unknown array name");
    if (!out) return (int)len;
    if ((size_t)cap < len) return fail(MPCX_E_INVALID, "buffer too small");
    if (hipMemcpy(out, src, len * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return fail(MPCX_E_DEVICE, "hipMemcpy failed");
    return (int)len;
}

int mpcx_lmpc_hetero_get_info(mpcx_lmpc_hetero_t f, int *count, int *active_words, int *m_ref, double *bytes_per_model)
{
    if (!f) return fail(MPCX_E_INVALID, "null bank");
    if (count) *count = f->count;
    if (active_words) *active_words = f->active_words;
    if (m_ref) *m_ref = f->m_ref;
    if (bytes_per_model) {
        const auto &D = f->dev0;
        // what a solve reads of its own model: -Hinv and G Hinv once (assemble), the model and weight arrays, the bounds
        *bytes_per_model = 8.0 * ((double)D.nz * D.ldy + (double)D.nx * D.nx + (double)D.nx * D.nu + (double)D.ny * D.nx +
                                  (double)(D.ph + 1) * (D.ny + D.nu) + (double)D.ph * D.nu + 2.0 * D.ldz + 2.0 * D.ldg);
    }
    return MPCX_OK;
}

int mpcx_lmpc_hetero_solve_batch(mpcx_lmpc_hetero_t f, const mpcx_lmpc_batch *b, const int32_t *model_index, void *stream)
{
    if (!f || !b) return fail(MPCX_E_INVALID, "null argument");
    if (hipSetDevice(f->device) != hipSuccess) return fail(MPCX_E_DEVICE, "hipSetDevice failed");
    if (b->batch == 0) return MPCX_OK;
    if (b->batch < 0 || !b->x0 || !b->u0 || !b->cmd) return fail(MPCX_E_INVALID, "x0, u0 and cmd are required");
    if (!model_index && b->batch != f->count) return fail(MPCX_E_INVALID, "without a model index the batch must be the bank: instance b uses controller b");
    const auto &d = f->d;
    mpcx::LmpcBatchDev B{};
    B.batch = b->batch; B.x0 = b->x0; B.u0 = b->u0;
    // references: per instance or per step from the caller; "shared" = each controller's own (its setReferences), read through the model
    auto refsel = [&](const double *p, int mode, int n, const double *&op, long &bs, long &ks) -> bool {
        if (mode == MPCX_REF_SHARED) { op = nullptr; bs = 0; ks = n; return true; }
        if (!p) return false;
        if (mode == MPCX_REF_PER_INSTANCE) { op = p; bs = n; ks = 0; return true; }
        if (mode == MPCX_REF_PER_STEP) { op = p; bs = (long)d.ph * n; ks = n; return true; }
        return false;
    };
    if (!refsel(b->yref, b->yref_mode, d.ny, B.yref, B.yref_bs, B.yref_ks) || !refsel(b->uref, b->uref_mode, d.nu, B.uref, B.uref_bs, B.uref_ks) ||
        !refsel(b->duref, b->duref_mode, d.nu, B.duref, B.duref_bs, B.duref_ks) || !refsel(b->dmeas, b->dmeas_mode, d.ndu, B.dmeas, B.dmeas_bs, B.dmeas_ks))
        return fail(MPCX_E_INVALID, "reference array missing for a non-shared mode, or unknown mode");
    B.cmd = b->cmd; B.cost = b->cost; B.status = b->status; B.solver_status = b->solver_status;
    B.is_feasible = b->is_feasible; B.iterations = b->iterations;
    B.active_lower = b->active_lower; B.active_upper = b->active_upper;
    B.seq_state = b->seq_state; B.seq_output = b->seq_output; B.seq_input = b->seq_input;
    B.polish_rounds = b->polish_rounds; B.active_count = b->active_count;
    B.warm_lower = b->warm_active_lower; B.warm_upper = b->warm_active_upper; B.warm_shift = b->warm_shift;
    if ((B.warm_lower == nullptr) != (B.warm_upper == nullptr)) return fail(MPCX_E_INVALID, "warm_active_lower and warm_active_upper go together");
    B.n_models = f->count; B.model_index = model_index;
    if ((size_t)b->batch > f->ws_cap) {
        if (f->ws) (void)hipFree(f->ws);
        f->ws = nullptr; f->ws_cap = 0;
        if (hipMalloc(reinterpret_cast<void **>(&f->ws), (size_t)b->batch * f->dev0.wsld * sizeof(double)) != hipSuccess)
            return fail(MPCX_E_DEVICE, "workspace allocation failed");
        f->ws_cap = (size_t)b->batch;
    }
    const int lr = mpcx::lmpc_launch(f->dev0, f->models_d, B, f->ws, stream, 7, -1);      // roll-out assemble, lean solve, ADMM fallback
    if (lr == -2) return fail(MPCX_E_UNSUPPORTED, "problem dimensions exceed the kernel's LDS budget");
    if (lr != 0) return fail(MPCX_E_DEVICE, std::string("kernel launch failed: ") + hipGetErrorString(hipGetLastError()));
    return MPCX_OK;
}

int mpcx_lmpc_hetero_time_solve_batch(mpcx_lmpc_hetero_t f, const mpcx_lmpc_batch *b, const int32_t *model_index, void *stream, int repeats, float *ms_mean)
{
    if (!f || !b || !ms_mean || repeats < 1) return fail(MPCX_E_INVALID, "bad argument");
    int rc = mpcx_lmpc_hetero_solve_batch(f, b, model_index, stream);
    if (rc != MPCX_OK) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(MPCX_E_DEVICE, "hipEventCreate failed");
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < repeats && rc == MPCX_OK; i++) rc = mpcx_lmpc_hetero_solve_batch(f, b, model_index, stream);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_mean = ms / (float)repeats;
    return rc;
}

/* profiling aid: mean time (ms) of the assemble and of the solve kernel, each timed alone with
 * HIP events on `stream` (the solve kernel re-reads the workspace the assemble kernel left). */
int mpcx_lmpc_debug_time_kernels(mpcx_lmpc_t h, const mpcx_lmpc_batch *b, void *stream, int repeats, float *ms2)
{
    CHECK_H(h);
    if (!b || !ms2 || repeats < 1 || h->host_only) return fail(MPCX_E_INVALID, "bad argument");
    int rc = mpcx_lmpc_solve_batch(h, b, stream);      // sizes the workspace, fills it
    if (rc != MPCX_OK) return rc;
    mpcx::LmpcBatchDev B;
    rc = make_batch(h, b, B);
    if (rc != MPCX_OK) return rc;
    int fast = -1;
    if (b->uref_mode == MPCX_REF_SHARED && b->duref_mode == MPCX_REF_SHARED && b->dmeas_mode == MPCX_REF_SHARED && !h->force_generic) {
        if (b->yref_mode == MPCX_REF_SHARED) fast = 0;
        else if (b->yref_mode == MPCX_REF_PER_INSTANCE) fast = 1;
    }
    if (fast >= 0 && h->dev.group_ok && (h->use_fused == 2 || (h->use_fused < 0 && (b->batch > h->total_batch ? b->batch : h->total_batch) <= h->group_max))) { B.fused = fast + 3; B.done = h->done; }
    else if (fast >= 0 && h->dev.fused_ok && !B.dbg_cycles && h->use_fused == 1) { B.fused = fast + 1; B.pcounter = h->pcounter; }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int which = 1; which <= 4; which *= 2) {
        float ms = 0;
        if (which == 2 && h->dev.cost_direct) {
            // with pending costs the solve leaves w in t0's place: every timed launch needs a freshly assembled workspace
            for (int i = 0; i < repeats; i++) {
                mpcx::lmpc_launch(h->dev, h->dev_d, B, h->ws, stream, 1, fast);
                (void)hipEventRecord(e0, s);
                mpcx::lmpc_launch(h->dev, h->dev_d, B, h->ws, stream, 2, fast);
                (void)hipEventRecord(e1, s);
                (void)hipEventSynchronize(e1);
                float one = 0;
                (void)hipEventElapsedTime(&one, e0, e1);
                ms += one;
            }
        } else {
            (void)hipEventRecord(e0, s);
            for (int i = 0; i < repeats; i++) mpcx::lmpc_launch(h->dev, h->dev_d, B, h->ws, stream, which, fast);
            (void)hipEventRecord(e1, s);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        ms2[which == 1 ? 0 : (which == 2 ? 1 : 2)] = ms / (float)repeats;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return MPCX_OK;
}

/* testing aid: how many full set-ups (condensing + device rebuild) and how many reference-only refreshes have run */
int mpcx_lmpc_debug_setup_counts(mpcx_lmpc_t h, int *full, int *refs)
{
    CHECK_H(h);
    if (full) *full = h->n_full_setups;
    if (refs) *refs = h->n_ref_refreshes;
    return MPCX_OK;
}

/* experiment / testing knob: 1 = wherever the dimensions allow, compute each instance's record inside the solve kernel instead of
 * handing it over through the HBM workspace; 0 = never; -1 = automatic (the default: from 16384 instances on) */
int mpcx_lmpc_debug_use_fused(mpcx_lmpc_t h, int on)
{
    CHECK_H(h);
    h->use_fused = on < 0 ? -1 : (on > 2 ? 2 : on);      // 2: lmpc_solve_group (assemble + solve in one workgroup of sixteen wavefronts)
    h->dirty = true;
    return MPCX_OK;
}

/* Sharding: this handle is given contiguous shards of a batch of `total` instances (one rank of N).  The kernel form is chosen for the
 * whole batch's size, so that a shard's results are, bit for bit, the rows of the unsharded solve also where the two sizes lie on either
 * side of the threshold between the in-workgroup form and the two-kernel form (0: every call is a whole batch, the default). */
int mpcx_lmpc_set_total_batch(mpcx_lmpc_t h, int total)
{
    CHECK_H(h);
    if (total < 0) return fail(MPCX_E_INVALID, "negative batch");
    h->total_batch = total;
    return MPCX_OK;
}

/* testing aid: 1 = always use the generic (roll-out) assemble kernel */
int mpcx_lmpc_debug_force_generic(mpcx_lmpc_t h, int on)
{
    CHECK_H(h);
    h->force_generic = on != 0;
    return MPCX_OK;
}

/* experiment knob: rounds the polish-only kernel may spend before handing an instance to the ADMM kernel, and the number of
 * ADMM iterations between two polish attempts there */
int mpcx_lmpc_debug_set_rounds(mpcx_lmpc_t h, int rounds0, int check_every)
{
    CHECK_H(h);
    h->dbg_rounds0 = rounds0; h->dbg_check_every = check_every;
    h->dirty = true;
    return MPCX_OK;
}

/* profiling aid: device buffer [B x 8] of int64 receiving per-phase cycle stamps */
int mpcx_lmpc_debug_set_cycle_buffer(mpcx_lmpc_t h, void *dev_ptr)
{
    CHECK_H(h);
    h->dbg_cycles = static_cast<long long *>(dev_ptr);
    return MPCX_OK;
}

/* ---- testing aid (not part of the reference-facing surface): copy a condensed array to
 * the host so that CPU-only tests can check the set-up without a GPU. ------------------ */
int mpcx_lmpc_debug_get(mpcx_lmpc_t h, const char *name, double *out, int cap)
{
    CHECK_H(h);
    int rc = mpcx_lmpc_setup(h);
    if (rc != MPCX_OK) return rc;
    const auto &o = h->cond;
    const std::vector<double> *v = nullptr;
    std::vector<double> tmp;
    std::string n(name ? name : "");
    if (n == "H") v = &o.H; else if (n == "Kinv") v = &o.Kinv; else if (n == "Gr") v = &o.Gr;
    else if (n == "Gc") v = &o.Gc; else if (n == "Y") v = &o.Y; else if (n == "lw") v = &o.lw;
    else if (n == "uw") v = &o.uw; else if (n == "rho_b") v = &o.rho_b; else if (n == "lg0") v = &o.lg0;
    else if (n == "ug0") v = &o.ug0; else if (n == "rho_g") v = &o.rho_g;
    else if (n == "dims") {
        tmp = {(double)o.nz, (double)o.mg, (double)o.ldz, (double)o.ldg, (double)o.ldy, (double)o.nf,
               (double)o.n_ref, (double)o.m_ref, (double)o.neq_ref, (double)o.fixed_rows.size(),
               (double)h->dev.lds_per_wave, (double)(o.h_regularised ? 1 : 0)};
        v = &tmp;
    } else if (n == "MA0") v = &o.MA[0]; else if (n == "MA1") v = &o.MA[1]; else if (n == "Ym") v = &o.Ym;
    else if (n == "MA0p" || n == "MA1p" || n == "Ymp") {      // the packed copies lmpc_solve_group / lmpc_assemble_mfma read (computed here as at upload)
        const std::vector<double> &src = n == "Ymp" ? o.Ym : o.MA[n == "MA1p" ? 1 : 0];
        const int rows = n == "Ymp" ? o.ldy16 : o.rowsA, K = n == "Ymp" ? o.nz16 : o.kin;
        if (!src.empty()) { tmp.resize(mpcx::lmpc_packed_len(rows, K)); mpcx::lmpc_pack_mfma_tiles(src.data(), rows, K, tmp.data()); }
        v = &tmp;
    } else if (n == "flags") {           // which forms of the solve this controller can take: cost from its definition (lmpc_cost_mfma follows lmpc_solve), one-workgroup form, fused mat-vec form
        tmp = {(double)h->dev.cost_direct, (double)h->dev.group_ok, (double)h->dev.fused_ok};
        v = &tmp;
    } else if (n == "dims_maps") {
        tmp = {(double)o.kin, (double)o.nxp, (double)o.nup, (double)o.nyp, (double)o.ione, (double)o.nz16, (double)o.mg16,
               (double)o.ns, (double)o.ns16, (double)o.kq16, (double)o.rowsA, (double)o.ldy16};
        v = &tmp;
    } else if (n == "g_refrow") { tmp.assign(o.g_refrow.begin(), o.g_refrow.end()); v = &tmp; }
    else if (n == "g_step") { tmp.assign(o.g_step.begin(), o.g_step.end()); v = &tmp; }
    else if (n == "g_kind") { tmp.assign(o.g_kind.begin(), o.g_kind.end()); v = &tmp; }
    else if (n == "g_comp") { tmp.assign(o.g_comp.begin(), o.g_comp.end()); v = &tmp; }
    else return fail(MPCX_E_INVALID, "unknown array name");
    if (!out) return (int)v->size();
    if (cap < (int)v->size()) return fail(MPCX_E_INVALID, "buffer too small");
    std::memcpy(out, v->data(), sizeof(double) * v->size());
    return (int)v->size();
}

}  // extern "C"
