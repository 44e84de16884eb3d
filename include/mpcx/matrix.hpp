// mpc::mat / mpc::cvec / mpc::rvec -- the small matrix types of the front-end headers.
//
// libmpc++ takes its matrix types from Eigen (reference include/mpc/Types.hpp:42-52), which is not part of this
// repository.  This header is a deliberately small stand-in with Eigen's spelling for what code written against the
// reference uses: controller set-up on the host (comma initialisation, setZero/Ones/Identity/Constant, Zero()/Ones(),
// array() +=/-=, *=, col(), isApprox, streaming) and the bodies of the NLMPC user hooks (element access, rows()/cols(),
// row()/col()/segment()/head()/tail()/transpose(), array().square()/abs(), sum()/squaredNorm()/norm()/dot(),
// element-wise + and -, scalar and matrix-vector products).  Storage is column-major doubles like Eigen's default.
//
// Two storage kinds:
//   * a dynamic dimension (mpc::Dynamic) -> heap storage, host only;
//   * fixed dimensions -> an inline array behind a small header, trivially copyable, usable in host AND device code.
//     A device lambda can therefore capture such a matrix by value, and the NLMPC engine hands its hooks *views*: the
//     header can point at a trajectory that lives in the wavefront's LDS slice (row-major, one row per horizon step)
//     with the finite-difference perturbation or the line-search step applied on the fly, or at a strided output
//     column in the workspace -- the hook reads X(i, j) and writes in_con(k) exactly as in the reference, and no lane
//     ever copies a trajectory.
#pragma once

#if defined(__HIPCC_RTC__)
// compiled at run time by hipRTC (device code only): no host headers, no heap-backed dynamic sizes, no streams
#define MPCX_HD __device__ __forceinline__
#define MPCX_HOST_API 0
#else
#include <cmath>
#include <initializer_list>
#include <iostream>
#include <stdexcept>
#include <vector>
#define MPCX_HOST_API 1
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MPCX_HD __host__ __device__ __forceinline__
#else
#define MPCX_HD inline
#endif
#endif

namespace mpc {

constexpr int Dynamic = -1;

template <int M = Dynamic, int N = Dynamic> class mat;

namespace detail {

// ---- lazy vector expressions (one index) ------------------------------------------------------------------------
template <class D> struct VecX {
    MPCX_HD const D &self() const { return static_cast<const D &>(*this); }
    MPCX_HD int size() const { return self().size(); }
    MPCX_HD int rows() const { return self().size(); }
    MPCX_HD double operator()(int i) const { return self().coeff(i); }
    MPCX_HD double operator[](int i) const { return self().coeff(i); }
    MPCX_HD double sum() const { double s = 0; for (int i = 0; i < size(); ++i) s += self().coeff(i); return s; }
    MPCX_HD double squaredNorm() const { double s = 0; for (int i = 0; i < size(); ++i) { const double v = self().coeff(i); s += v * v; } return s; }
    MPCX_HD double norm() const { return sqrt(squaredNorm()); }
    MPCX_HD double maxCoeff() const { double m = self().coeff(0); for (int i = 1; i < size(); ++i) { const double v = self().coeff(i); m = v > m ? v : m; } return m; }
    MPCX_HD double minCoeff() const { double m = self().coeff(0); for (int i = 1; i < size(); ++i) { const double v = self().coeff(i); m = v < m ? v : m; } return m; }
    template <class E> MPCX_HD double dot(const VecX<E> &o) const { double s = 0; for (int i = 0; i < size(); ++i) s += self().coeff(i) * o.self().coeff(i); return s; }
    MPCX_HD const D &transpose() const { return self(); }         // one index: orientation carries no information here
    MPCX_HD const D &array() const { return self(); }
    MPCX_HD const D &matrix() const { return self(); }
    MPCX_HD const D &eval() const { return self(); }
    MPCX_HD auto square() const;
    MPCX_HD auto abs() const;
    MPCX_HD auto segment(int start, int n) const;
    MPCX_HD auto head(int n) const;
    MPCX_HD auto tail(int n) const;
};
template <class A, int OP> struct Unary : VecX<Unary<A, OP>> {
    A a;
    MPCX_HD explicit Unary(const A &a_) : a(a_) {}
    MPCX_HD int size() const { return a.size(); }
    MPCX_HD double coeff(int i) const { const double v = a.coeff(i); return OP == 0 ? v * v : (OP == 1 ? fabs(v) : -v); }
};
template <class A> struct Seg : VecX<Seg<A>> {
    A a; int s, n;
    MPCX_HD Seg(const A &a_, int s_, int n_) : a(a_), s(s_), n(n_) {}
    MPCX_HD int size() const { return n; }
    MPCX_HD double coeff(int i) const { return a.coeff(s + i); }
};
template <class A, class B, int OP> struct Binary : VecX<Binary<A, B, OP>> {
    A a; B b;
    MPCX_HD Binary(const A &a_, const B &b_) : a(a_), b(b_) {}
    MPCX_HD int size() const { return a.size(); }
    MPCX_HD double coeff(int i) const { return OP == 0 ? a.coeff(i) + b.coeff(i) : (OP == 1 ? a.coeff(i) - b.coeff(i) : a.coeff(i) * b.coeff(i)); }
};
template <class A> struct Scaled : VecX<Scaled<A>> {
    A a; double s;
    MPCX_HD Scaled(const A &a_, double s_) : a(a_), s(s_) {}
    MPCX_HD int size() const { return a.size(); }
    MPCX_HD double coeff(int i) const { return s * a.coeff(i); }
};
template <class D> MPCX_HD auto VecX<D>::square() const { return Unary<D, 0>(self()); }
template <class D> MPCX_HD auto VecX<D>::abs() const { return Unary<D, 1>(self()); }
template <class D> MPCX_HD auto VecX<D>::segment(int start, int n) const { return Seg<D>(self(), start, n); }
template <class D> MPCX_HD auto VecX<D>::head(int n) const { return Seg<D>(self(), 0, n); }
template <class D> MPCX_HD auto VecX<D>::tail(int n) const { return Seg<D>(self(), size() - n, n); }
template <class A, class B> MPCX_HD auto operator+(const VecX<A> &a, const VecX<B> &b) { return Binary<A, B, 0>(a.self(), b.self()); }
template <class A, class B> MPCX_HD auto operator-(const VecX<A> &a, const VecX<B> &b) { return Binary<A, B, 1>(a.self(), b.self()); }
template <class A> MPCX_HD auto operator-(const VecX<A> &a) { return Unary<A, 2>(a.self()); }
template <class A> MPCX_HD auto operator*(double s, const VecX<A> &a) { return Scaled<A>(a.self(), s); }
template <class A> MPCX_HD auto operator*(const VecX<A> &a, double s) { return Scaled<A>(a.self(), s); }
template <class A> MPCX_HD auto operator/(const VecX<A> &a, double s) { return Scaled<A>(a.self(), 1.0 / s); }

// what a view header says about the storage behind a fixed-size matrix
enum : int { kOwn = 0, kTraj = 1, kZero = 2, kOut = 3 };
struct ViewHeader {
    const double *v = nullptr;       // kTraj: trajectory, row-major [rows x cols] (row = horizon step)
    const double *d = nullptr;       //        optional direction: value = v + al * d (a trial point of the line search)
    const double *o1 = nullptr, *o2 = nullptr;   // optional replacement rows for r1 / r2 (outputs of a perturbed step)
    double *w = nullptr;             // kOut: element k lives at w[k * ws]
    double al = 0.0, dp = 0.0;       // step length; perturbation added to elements (r1, c) and (r2, c)
    int mode = kOwn, r1 = -1, r2 = -1, c = -1, ws = 1, pad_ = 0;
};

}  // namespace detail

#if MPCX_HOST_API
// ---------------------------------------------------------------------------------------------------------------
// dynamic sizes: heap storage, host only (set-up code with run-time dimensions)
// ---------------------------------------------------------------------------------------------------------------
template <int M, int N>
class mat {
    int r_ = M < 0 ? 0 : M, c_ = N < 0 ? 0 : N;
    std::vector<double> a_;

public:
    class CommaInit {
        mat &m_;
        int k_ = 0;

    public:
        CommaInit(mat &m, double first) : m_(m) { put(first); }
        CommaInit &operator,(double v) { put(v); return *this; }
        void put(double v)
        {
            if (k_ >= m_.r_ * m_.c_) throw std::out_of_range("too many coefficients passed to comma initialiser");
            const int i = k_ / m_.c_, j = k_ % m_.c_;      // row-major fill order, as Eigen
            m_(i, j) = v;
            ++k_;
        }
    };
    struct ArrayProxy {
        mat &m;
        ArrayProxy &operator-=(double s) { for (double &v : m.a_) v -= s; return *this; }
        ArrayProxy &operator+=(double s) { for (double &v : m.a_) v += s; return *this; }
        ArrayProxy &operator*=(double s) { for (double &v : m.a_) v *= s; return *this; }
    };
    struct ColProxy {
        mat &m;
        int j;
        template <int R2, int C2> ColProxy &operator=(const mat<R2, C2> &v)
        {
            if (v.size() != m.rows()) throw std::invalid_argument("column size mismatch");
            for (int i = 0; i < m.rows(); ++i) m(i, j) = v.data()[i];
            return *this;
        }
        CommaInit operator<<(double first) = delete;
    };

    mat() : a_((size_t)r_ * c_, 0.0) {}
    mat(int r, int c) : r_(r), c_(c), a_((size_t)r * c, 0.0) {}
    explicit mat(int n) : r_(N == 1 ? n : (M == 1 ? 1 : n)), c_(N == 1 ? 1 : (M == 1 ? n : 1)), a_((size_t)n, 0.0) {}

    void resize(int r, int c) { r_ = r; c_ = c; a_.assign((size_t)r * c, 0.0); }
    void resize(int n) { if (c_ == 1 || N == 1) resize(n, 1); else resize(1, n); }
    int rows() const { return r_; }
    int cols() const { return c_; }
    int size() const { return r_ * c_; }
    double *data() { return a_.data(); }
    const double *data() const { return a_.data(); }
    double &operator()(int i, int j) { return a_[(size_t)i + (size_t)j * r_]; }
    double operator()(int i, int j) const { return a_[(size_t)i + (size_t)j * r_]; }
    double &operator()(int i) { return a_[(size_t)i]; }
    double operator()(int i) const { return a_[(size_t)i]; }
    double &operator[](int i) { return a_[(size_t)i]; }
    double operator[](int i) const { return a_[(size_t)i]; }

    CommaInit operator<<(double first) { return CommaInit(*this, first); }
    mat &setZero() { return setConstant(0.0); }
    mat &setOnes() { return setConstant(1.0); }
    mat &setConstant(double v) { for (double &x : a_) x = v; return *this; }
    mat &setIdentity()
    {
        setZero();
        for (int i = 0; i < (r_ < c_ ? r_ : c_); ++i) (*this)(i, i) = 1.0;
        return *this;
    }
    mat &fill(double v) { return setConstant(v); }
    static mat Zero() { mat m; return m; }
    static mat Zero(int r, int c) { return mat(r, c); }
    static mat Ones() { mat m; m.setOnes(); return m; }
    static mat Identity() { mat m; m.setIdentity(); return m; }
    ArrayProxy array() { return ArrayProxy{*this}; }
    ColProxy col(int j) { return ColProxy{*this, j}; }
    mat &operator*=(double s) { for (double &v : a_) v *= s; return *this; }
    template <int R2, int C2> mat &operator+=(const mat<R2, C2> &o) { for (int i = 0; i < size(); ++i) a_[i] += o.data()[i]; return *this; }
    bool isApprox(const mat &o, double prec = 1e-12) const
    {
        // Eigen's definition: ||a - b||_2 <= prec * min(||a||_2, ||b||_2)
        if (o.size() != size()) return false;
        double d = 0, na = 0, nb = 0;
        for (int i = 0; i < size(); ++i) {
            d += (a_[i] - o.a_[i]) * (a_[i] - o.a_[i]); na += a_[i] * a_[i]; nb += o.a_[i] * o.a_[i];
        }
        return std::sqrt(d) <= prec * std::sqrt(na < nb ? na : nb);
    }
    friend std::ostream &operator<<(std::ostream &os, const mat &m)
    {
        for (int i = 0; i < m.r_; ++i) {
            for (int j = 0; j < m.c_; ++j) os << (j ? " " : "") << m(i, j);
            if (i + 1 < m.r_) os << "\n";
        }
        return os;
    }
};

#endif   // MPCX_HOST_API

// ---------------------------------------------------------------------------------------------------------------
// fixed sizes: view header + inline storage, host and device
// ---------------------------------------------------------------------------------------------------------------
namespace detail {
template <class Mt> struct RowX : VecX<RowX<Mt>> {
    const Mt *m; int i;
    MPCX_HD RowX(const Mt *m_, int i_) : m(m_), i(i_) {}
    MPCX_HD int size() const { return Mt::Cols; }
    MPCX_HD double coeff(int j) const { return (*m)(i, j); }
};
template <class Mt> struct ColX : VecX<ColX<Mt>> {
    const Mt *m; int j;
    MPCX_HD ColX(const Mt *m_, int j_) : m(m_), j(j_) {}
    MPCX_HD int size() const { return Mt::Rows; }
    MPCX_HD double coeff(int i) const { return (*m)(i, j); }
};
template <class Mt> struct FlatX : VecX<FlatX<Mt>> {           // all coefficients, column-major order (Eigen's linear index)
    const Mt *m;
    MPCX_HD explicit FlatX(const Mt *m_) : m(m_) {}
    MPCX_HD int size() const { return Mt::Rows * Mt::Cols; }
    MPCX_HD double coeff(int k) const { return (*m)(k % Mt::Rows, k / Mt::Rows); }
};
template <class Mt, class V> struct MatVec : VecX<MatVec<Mt, V>> {
    const Mt *m; V v;
    MPCX_HD MatVec(const Mt *m_, const V &v_) : m(m_), v(v_) {}
    MPCX_HD int size() const { return Mt::Rows; }
    MPCX_HD double coeff(int i) const { double s = 0; for (int j = 0; j < Mt::Cols; ++j) s += (*m)(i, j) * v.coeff(j); return s; }
};
}  // namespace detail

template <int M, int N>
    requires(M >= 0 && N >= 0)
class mat<M, N> : public detail::VecX<mat<M, N>> {
    detail::ViewHeader h_;
    double a_[M * N > 0 ? M * N : 1];

public:
    static constexpr int Rows = M, Cols = N;
    // ---- construction -----------------------------------------------------------------------------------
    MPCX_HD mat() { for (int k = 0; k < M * N; ++k) a_[k] = 0.0; }
    MPCX_HD mat(int, int) : mat() {}                              // fixed sizes given again at run time (reference test code does)
    MPCX_HD explicit mat(int) : mat() {}
    template <class E> MPCX_HD mat(const detail::VecX<E> &e) { for (int k = 0; k < M * N; ++k) a_[k] = e.self().coeff(k); }
    template <class E> MPCX_HD mat &operator=(const detail::VecX<E> &e)
    {
        for (int k = 0; k < M * N; ++k) (*this)(k) = e.self().coeff(k);
        return *this;
    }
    // views (what the NLMPC engine hands to the user hooks; see the header comment)
    MPCX_HD static mat trajectory(const double *rows, const double *dir = nullptr, double al = 0.0)
    {
        mat m(detail::ViewHeader{}); m.h_.mode = detail::kTraj; m.h_.v = rows; m.h_.d = dir; m.h_.al = al; return m;
    }
    MPCX_HD static mat zeros_view() { mat m(detail::ViewHeader{}); m.h_.mode = detail::kZero; return m; }
    MPCX_HD static mat output(double *w, int stride) { mat m(detail::ViewHeader{}); m.h_.mode = detail::kOut; m.h_.w = w; m.h_.ws = stride; return m; }
    MPCX_HD mat &perturb(int r1, int r2, int c, double dp) { h_.r1 = r1; h_.r2 = r2; h_.c = c; h_.dp = dp; return *this; }
    MPCX_HD mat &replace_rows(const double *o1, const double *o2) { h_.o1 = o1; h_.o2 = o2; return *this; }

    // ---- shape and element access -------------------------------------------------------------------------
    MPCX_HD void resize(int, int) {}
    MPCX_HD void resize(int) {}
    MPCX_HD static constexpr int rows() { return M; }
    MPCX_HD static constexpr int cols() { return N; }
    MPCX_HD static constexpr int size() { return M * N; }
    MPCX_HD double *data() { return a_; }
    MPCX_HD const double *data() const { return a_; }
    MPCX_HD double operator()(int i, int j) const
    {
        if (h_.mode == detail::kOwn) return a_[i + j * M];
        if (h_.mode == detail::kTraj) {
            if (h_.o1 && i == h_.r1) return h_.o1[j];
            if (h_.o2 && i == h_.r2) return h_.o2[j];
            double x = h_.v[i * N + j];
            if (h_.d) x += h_.al * h_.d[i * N + j];
            if (j == h_.c && (i == h_.r1 || i == h_.r2)) x += h_.dp;
            return x;
        }
        if (h_.mode == detail::kOut) return h_.w[(i + j * M) * h_.ws];
        return 0.0;
    }
    MPCX_HD double &operator()(int i, int j) { return h_.mode == detail::kOut ? h_.w[(i + j * M) * h_.ws] : a_[i + j * M]; }
    MPCX_HD double operator()(int k) const { return (*this)(k % M, k / M); }
    MPCX_HD double &operator()(int k) { return h_.mode == detail::kOut ? h_.w[k * h_.ws] : a_[k]; }
    MPCX_HD double operator[](int k) const { return (*this)(k); }
    MPCX_HD double &operator[](int k) { return (*this)(k); }
    MPCX_HD double coeff(int k) const { return (*this)(k); }
    MPCX_HD double &coeffRef(int k) { return (*this)(k); }

    // ---- expressions ------------------------------------------------------------------------------------------
    MPCX_HD auto row(int i) const { return detail::RowX<mat>(this, i); }
    MPCX_HD auto col(int j) const { return detail::ColX<mat>(this, j); }
    MPCX_HD auto array() const { return detail::FlatX<mat>(this); }
    template <class E> MPCX_HD auto operator*(const detail::VecX<E> &v) const { return detail::MatVec<mat, E>(this, v.self()); }
    template <class E> MPCX_HD mat &operator+=(const detail::VecX<E> &e) { for (int k = 0; k < M * N; ++k) (*this)(k) += e.self().coeff(k); return *this; }
    template <class E> MPCX_HD mat &operator-=(const detail::VecX<E> &e) { for (int k = 0; k < M * N; ++k) (*this)(k) -= e.self().coeff(k); return *this; }
    MPCX_HD mat &operator*=(double s) { for (int k = 0; k < M * N; ++k) (*this)(k) *= s; return *this; }
    MPCX_HD mat &setZero() { return setConstant(0.0); }
    MPCX_HD mat &setOnes() { return setConstant(1.0); }
    MPCX_HD mat &setConstant(double v) { for (int k = 0; k < M * N; ++k) (*this)(k) = v; return *this; }
    MPCX_HD mat &fill(double v) { return setConstant(v); }
    MPCX_HD mat &setIdentity()
    {
        setZero();
        for (int i = 0; i < (M < N ? M : N); ++i) (*this)(i, i) = 1.0;
        return *this;
    }
    MPCX_HD static mat Zero() { return mat(); }
    MPCX_HD static mat Zero(int, int) { return mat(); }
    MPCX_HD static mat Ones() { mat m; m.setOnes(); return m; }
    MPCX_HD static mat Identity() { mat m; m.setIdentity(); return m; }

    // ---- host conveniences of set-up code -------------------------------------------------------------------------
    class CommaInit {
        mat &m_;
        int k_ = 0;

    public:
        MPCX_HD CommaInit(mat &m, double first) : m_(m) { put(first); }
        MPCX_HD CommaInit &operator,(double v) { put(v); return *this; }
        MPCX_HD void put(double v)
        {
            if (k_ < M * N) m_(k_ / N, k_ % N) = v;                  // row-major fill order, as Eigen
#if MPCX_HOST_API && !defined(__HIP_DEVICE_COMPILE__)
            else throw std::out_of_range("too many coefficients passed to comma initialiser");
#endif
            ++k_;
        }
    };
    struct ArrayProxy : detail::VecX<ArrayProxy> {
        mat &m;
        MPCX_HD explicit ArrayProxy(mat &m_) : m(m_) {}
        MPCX_HD int size() const { return M * N; }
        MPCX_HD double coeff(int k) const { return static_cast<const mat &>(m)(k); }
        MPCX_HD ArrayProxy &operator-=(double s) { for (int k = 0; k < M * N; ++k) m(k) -= s; return *this; }
        MPCX_HD ArrayProxy &operator+=(double s) { for (int k = 0; k < M * N; ++k) m(k) += s; return *this; }
        MPCX_HD ArrayProxy &operator*=(double s) { for (int k = 0; k < M * N; ++k) m(k) *= s; return *this; }
    };
    struct ColProxy : detail::VecX<ColProxy> {
        mat &m;
        int j;
        MPCX_HD ColProxy(mat &m_, int j_) : m(m_), j(j_) {}
        MPCX_HD int size() const { return M; }
        MPCX_HD double coeff(int i) const { return static_cast<const mat &>(m)(i, j); }
        template <class E> MPCX_HD ColProxy &operator=(const detail::VecX<E> &v) { for (int i = 0; i < M; ++i) m(i, j) = v.self().coeff(i); return *this; }
#if MPCX_HOST_API
        template <int R2, int C2> ColProxy &operator=(const mat<R2, C2> &v) requires(R2 < 0 || C2 < 0)
        {
            for (int i = 0; i < M; ++i) m(i, j) = v.data()[i];
            return *this;
        }
#endif
    };
    MPCX_HD CommaInit operator<<(double first) { return CommaInit(*this, first); }
    MPCX_HD ArrayProxy array() { return ArrayProxy(*this); }
    MPCX_HD ColProxy col(int j) { return ColProxy(*this, j); }
#if MPCX_HOST_API
    bool isApprox(const mat &o, double prec = 1e-12) const
    {
        double d = 0, na = 0, nb = 0;
        for (int k = 0; k < M * N; ++k) {
            const double a = (*this)(k), b = o(k);
            d += (a - b) * (a - b); na += a * a; nb += b * b;
        }
        return std::sqrt(d) <= prec * std::sqrt(na < nb ? na : nb);
    }
    friend std::ostream &operator<<(std::ostream &os, const mat &m)
    {
        for (int i = 0; i < M; ++i) {
            for (int j = 0; j < N; ++j) os << (j ? " " : "") << m(i, j);
            if (i + 1 < M) os << "\n";
        }
        return os;
    }
#endif

private:
    MPCX_HD explicit mat(const detail::ViewHeader &h) : h_(h) {}     // a view: the inline array is never touched
};

template <int N = Dynamic> using cvec = mat<N, 1>;
template <int N = Dynamic> using rvec = mat<1, N>;

}  // namespace mpc
