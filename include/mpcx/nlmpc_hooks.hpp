// User hooks with the reference's own signatures as a model of the NLMPC engine.
//
// libmpc++ takes five closures (IDimensionable.hpp:94-149, set through NLMPC::setStateSpaceFunction, setOutputFunction,
// setObjectiveFunction, setIneqConFunction, setEqConFunction, NLMPC.hpp:139-281):
//     StateFunHandle  void(cvec<nx>& dx, const cvec<nx>& x, const cvec<nu>& u, const unsigned int& step)
//     OutFunHandle    void(cvec<ny>& y,  const cvec<nx>& x, const cvec<nu>& u, const unsigned int& step)
//     ObjFunHandle    double(const mat<ph+1,nx>& X, const mat<ph+1,ny>& Y, const mat<ph+1,nu>& U, const double& slack)
//     IConFunHandle   void(cvec<ineq>& g, const mat<ph+1,nx>& X, const mat<ph+1,ny>& Y, const mat<ph+1,nu>& U, const double& slack)
//     EConFunHandle   void(cvec<eq>& h, const mat<ph+1,nx>& X, const mat<ph+1,nu>& U)
// HookModel adapts a set of callables with exactly these parameter lists -- device lambdas, functor structs -- to the
// engine (mpcx/nlmpc_engine.hpp): the matrix arguments are mpc::mat views of the trajectory in the wavefront's LDS slice
// with the finite-difference perturbation or the line-search step applied on the fly (mpcx/matrix.hpp), the output
// vectors are strided views of the workspace.  The callables travel to the device as bytes (NlmpcDev::params), so what
// they capture must be captured by value and be trivially copyable (numbers, mpc::mat / mpc::cvec of fixed size).
//
// Two hook containers:
//   HookSet<F...>     the callables' types are known together: every call is inlined (setHooks(), run-time compiled
//                     sources);
//   ErasedHooks<...>  the callables arrive one setter at a time, as in the reference's API, so their types are never known
//                     together: each is reached through a device function pointer taken when its setter runs.
#pragma once

#include "matrix.hpp"
#include "nlmpc_device.hpp"
#include "nlmpc_engine.hpp"

namespace mpcx {

struct NoHook {};      // an unset hook: no outputs / zero cost / no rows

template <class FDyn, class FObj, class FIneq = NoHook, class FEq = NoHook, class FOut = NoHook>
struct HookSet {
    static constexpr bool kErased = false;
    FDyn fdyn; FObj fobj; FIneq fineq; FEq feq; FOut fout;
    template <class VX, class VU> __device__ void f(VX &dx, const VX &x, const VU &u, const unsigned &s) const { fdyn(dx, x, u, s); }
    template <class MX, class MY, class MU> __device__ double obj(const MX &X, const MY &Y, const MU &U, const double &e) const
    {
        if constexpr (__is_same(FObj, NoHook)) return 0.0; else return fobj(X, Y, U, e);
    }
    template <class VI, class MX, class MY, class MU> __device__ void ineq(VI &g, const MX &X, const MY &Y, const MU &U, const double &e) const
    {
        if constexpr (!__is_same(FIneq, NoHook)) fineq(g, X, Y, U, e);
    }
    template <class VE, class MX, class MU> __device__ void eq(VE &h, const MX &X, const MU &U) const
    {
        if constexpr (!__is_same(FEq, NoHook)) feq(h, X, U);
    }
    template <class VY, class VX, class VU> __device__ void out(VY &y, const VX &x, const VU &u, const unsigned &s) const
    {
        if constexpr (!__is_same(FOut, NoHook)) fout(y, x, u, s);
    }
};

template <int NX_, int NU_, int NY_, int PH_, int CH_, int NI_, int NE_, class Hooks>
struct HookModel {
    static constexpr bool VECTOR_HOOKS = true;
    static constexpr bool CONTINUOUS = true;            // "may be": NlmpcDev::continuous decides at run time
    static constexpr bool HAS_OUTPUT = true;            // "may have": NlmpcDev::has_output decides at run time
    static constexpr bool INEQ_USES_SLACK = true;
    static constexpr int NX = NX_, NU = NU_, NY = NY_, PH = PH_, CH = CH_, NI = NI_, NE = NE_;
    using MatX = mpc::mat<PH + 1, NX>;
    using MatU = mpc::mat<PH + 1, NU>;
    using MatY = mpc::mat<PH + 1, NY>;
    using VecX = mpc::cvec<NX>;
    using VecU = mpc::cvec<NU>;
    using VecY = mpc::cvec<NY>;
    using VecI = mpc::cvec<NI>;
    using VecE = mpc::cvec<NE>;
    __host__ __device__ static int nineq(int) { return NI; }
    __host__ __device__ static int neq_user(int) { return NE; }
    __device__ static const Hooks &H(const double *prm) { return *reinterpret_cast<const Hooks *>(prm); }

    __device__ static void f(double *dx, const double *x, const double *u, const double *prm, unsigned step)
    {
        VecX xv, dxv; VecU uv;
        for (int a = 0; a < NX; ++a) xv(a) = x[a];
        for (int a = 0; a < NU; ++a) uv(a) = u[a];
        const unsigned st = step;
        H(prm).f(dxv, xv, uv, st);
        for (int a = 0; a < NX; ++a) dx[a] = static_cast<const VecX &>(dxv)(a);
    }
    __device__ static void out(double *y, const double *x, const double *u, const double *prm, unsigned step)
    {
        VecX xv; VecU uv; VecY yv;
        for (int a = 0; a < NX; ++a) xv(a) = x[a];
        for (int a = 0; a < NU; ++a) uv(a) = u[a];
        const unsigned st = step;
        H(prm).out(yv, xv, uv, st);
        for (int a = 0; a < NY; ++a) y[a] = static_cast<const VecY &>(yv)(a);
    }
    __device__ static double cost(const MatX &X, const MatY &Y, const MatU &U, double e, const double *prm) { return H(prm).obj(X, Y, U, e); }
    __device__ static void ineq_all(VecI &g, const MatX &X, const MatY &Y, const MatU &U, double e, const double *prm) { H(prm).ineq(g, X, Y, U, e); }
    __device__ static void eq_all(VecE &h, const MatX &X, const MatU &U, const double *prm) { H(prm).eq(h, X, U); }
    // no declared structure: every constraint may read every state
    __device__ static bool ineq_reads_x(int, int) { return true; }
    __device__ static bool eq_reads_x(int, int) { return true; }
};

#if !defined(__HIPCC_RTC__)
// ---- hooks set one at a time: device function pointers ------------------------------------------------------------------
constexpr int kHookClosureBytes = 192;

template <int NX, int NU, int NY, int PH, int CH, int NI, int NE>
struct ErasedHooks {
    static constexpr bool kErased = true;
    using MatX = mpc::mat<PH + 1, NX>;
    using MatU = mpc::mat<PH + 1, NU>;
    using MatY = mpc::mat<PH + 1, NY>;
    using VecX = mpc::cvec<NX>;
    using VecU = mpc::cvec<NU>;
    using VecY = mpc::cvec<NY>;
    using VecI = mpc::cvec<NI>;
    using VecE = mpc::cvec<NE>;
    typedef void (*dyn_fn)(const void *, VecX &, const VecX &, const VecU &, const unsigned &);
    typedef void (*out_fn)(const void *, VecY &, const VecX &, const VecU &, const unsigned &);
    typedef double (*obj_fn)(const void *, const MatX &, const MatY &, const MatU &, const double &);
    typedef void (*ineq_fn)(const void *, VecI &, const MatX &, const MatY &, const MatU &, const double &);
    typedef void (*eq_fn)(const void *, VecE &, const MatX &, const MatU &);
    dyn_fn pdyn = nullptr; out_fn pout = nullptr; obj_fn pobj = nullptr; ineq_fn pineq = nullptr; eq_fn peq = nullptr;
    alignas(16) unsigned char cdyn[kHookClosureBytes], cout_[kHookClosureBytes], cobj[kHookClosureBytes], cineq[kHookClosureBytes],
        ceq[kHookClosureBytes];
    __device__ void f(VecX &dx, const VecX &x, const VecU &u, const unsigned &s) const { pdyn(cdyn, dx, x, u, s); }
    __device__ void out(VecY &y, const VecX &x, const VecU &u, const unsigned &s) const { if (pout) pout(cout_, y, x, u, s); }
    __device__ double obj(const MatX &X, const MatY &Y, const MatU &U, const double &e) const { return pobj ? pobj(cobj, X, Y, U, e) : 0.0; }
    __device__ void ineq(VecI &g, const MatX &X, const MatY &Y, const MatU &U, const double &e) const { if (pineq) pineq(cineq, g, X, Y, U, e); }
    __device__ void eq(VecE &h, const MatX &X, const MatU &U) const { if (peq) peq(ceq, h, X, U); }
};

namespace hookdetail {
// the device function a pointer is taken of: calls the closure at `c` with the hook's arguments
template <class F, class R, class... A> __device__ R trampoline(const void *c, A... a) { return (*static_cast<const F *>(c))(a...); }
template <class F, class R, class... A> __global__ void fetch_trampoline(R (**out)(const void *, A...)) { *out = &trampoline<F, R, A...>; }

// device address of trampoline<F, ...> for the signature of `slot`
template <class F, class R, class... A>
inline bool resolve(R (*&slot)(const void *, A...))
{
    R (**d)(const void *, A...) = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&d), sizeof(void *)) != hipSuccess) return false;
    hipLaunchKernelGGL((fetch_trampoline<F, R, A...>), dim3(1), dim3(1), 0, nullptr, d);
    const bool ok = hipMemcpy(&slot, d, sizeof(void *), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    return ok && slot != nullptr;
}
template <class F> inline void stash(unsigned char (&dst)[kHookClosureBytes], const F &f)
{
    static_assert(sizeof(F) <= kHookClosureBytes, "the hook captures too much: keep its captures within 192 bytes");
    static_assert(__is_trivially_copyable(F), "the hook must capture trivially copyable values (by value)");
    __builtin_memcpy(dst, &f, sizeof(F));
}
}  // namespace hookdetail
#endif   // !__HIPCC_RTC__

}  // namespace mpcx
