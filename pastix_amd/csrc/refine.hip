// refine.hip -- iterative refinement on the device: the Krylov vectors, the sparse matrix-vector product
// and the preconditioner (the triangular solves on the device-resident factors) all live on the GPU; the host sees
// scalars only.
//
// Reference: pastix_task_raff (src/sopalin/src/pastix.c:4300-4500) picks the refiner from IPARM_REFINEMENT
// (api.h:353-365): GMRES (raff_gmres.c), conjugate gradient (raff_grad.c), simple iterative refinement
// (raff_pivot.c), BiCGStab (raff_bicgstab.c).  All of them iterate on  b - A x  with the factorization as the
// preconditioner (API_CALL(up_down_smp) per application) and stop at ||b - A x|| / ||b|| < DPARM_EPSILON_REFINEMENT or
// after IPARM_ITERMAX iterations; IPARM_NBITER and DPARM_RELATIVE_ERROR report what happened.
//
// The matrix arrives as the caller's CSC (1-based; lower triangle for symmetric / Hermitian input) and is expanded once
// into a full CSR on the device, so that y = A x is one thread per row without atomics (deterministic).  Vectors stay
// in the caller's numbering; the preconditioner scatters into the factor's numbering (perm), runs
// pastix_amd_solve_device and gathers back.
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstring>
#include <new>
#include <vector>

#include "engine.h"

namespace {

struct zc { double x, y; };
__host__ __device__ inline double mulT(double a, double b) { return a * b; }
__host__ __device__ inline zc mulT(zc a, zc b) { return zc{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__host__ __device__ inline double cmulT(double a, double b) { return a * b; }                       // conj(a) * b
__host__ __device__ inline zc cmulT(zc a, zc b) { return zc{a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x}; }
__host__ __device__ inline double addT(double a, double b) { return a + b; }
__host__ __device__ inline zc addT(zc a, zc b) { return zc{a.x + b.x, a.y + b.y}; }
__host__ __device__ inline double zeroT(double) { return 0.0; }
__host__ __device__ inline zc zeroT(zc) { return zc{0.0, 0.0}; }

template <class T>
__global__ void k_spmv(int64_t n, const int64_t* __restrict__ rp, const int32_t* __restrict__ ci,
                       const T* __restrict__ v, const T* __restrict__ x, T* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T s = zeroT(T{});
  for (int64_t q = rp[i]; q < rp[i + 1]; q++) s = addT(s, mulT(v[q], x[ci[q]]));
  y[i] = s;
}
template <class T>
__global__ void k_scatter_perm(int64_t n, const int64_t* __restrict__ perm, const T* __restrict__ in, T* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[perm[i]] = in[i];
}
template <class T>
__global__ void k_gather_perm(int64_t n, const int64_t* __restrict__ perm, const T* __restrict__ in, T* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[perm[i]];
}
// y = a x + b y   (b == 0: y is not read)
template <class T>
__global__ void k_axpby(int64_t n, T a, const T* __restrict__ x, T b, T* __restrict__ y, int yread) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const T ax = mulT(a, x[i]);
  y[i] = yread ? addT(ax, mulT(b, y[i])) : ax;
}
constexpr int DOT_BLOCKS = 256;
// partial[b] = sum over the block's share of conj(u) * w; the host adds the DOT_BLOCKS partials in order
template <class T>
__global__ void k_dot(int64_t n, const T* __restrict__ u, const T* __restrict__ w, T* __restrict__ partial) {
  __shared__ T red[256];
  T s = zeroT(T{});
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    s = addT(s, cmulT(u[i], w[i]));
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] = addT(red[threadIdx.x], red[threadIdx.x + st]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

template <class T> struct HostT;
template <> struct HostT<double> { using type = double; };
template <> struct HostT<zc> { using type = std::complex<double>; };

template <class T>
struct Dev {
  using H = typename HostT<T>::type;
  pastix_amd_plan_t* p;
  hipStream_t s;
  int64_t n;
  int64_t* rp = nullptr; int32_t* ci = nullptr; T* val = nullptr; int64_t* perm = nullptr;
  T* partial = nullptr; T* xp = nullptr;
  std::vector<T*> owned;
  int rc = 0;
  static T toT(H h) { T t; std::memcpy(&t, &h, sizeof(T)); return t; }
  static H toH(T t) { H h; std::memcpy(&h, &t, sizeof(T)); return h; }
  T* vec() {
    T* v = nullptr;
    if (hipMalloc((void**)&v, (size_t)n * sizeof(T)) != hipSuccess) { rc = PASTIX_AMD_ERR_ALLOC; return nullptr; }
    owned.push_back(v);
    return v;
  }
  ~Dev() {
    for (T* v : owned) (void)hipFree(v);
    (void)hipFree(rp); (void)hipFree(ci); (void)hipFree(val); (void)hipFree(perm); (void)hipFree(partial);
  }
  dim3 grid() const { return dim3((unsigned)((n + 255) / 256)); }
  void spmv(const T* x, T* y) { hipLaunchKernelGGL(k_spmv<T>, grid(), dim3(256), 0, s, n, rp, ci, val, x, y); }
  void axpby(H a, const T* x, H b, T* y) {
    hipLaunchKernelGGL(k_axpby<T>, grid(), dim3(256), 0, s, n, toT(a), x, toT(b), y, b != H(0.0) ? 1 : 0);
  }
  void copy(const T* x, T* y) { (void)hipMemcpyAsync(y, x, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, s); }
  // y = 0 whatever y or anything else holds (0 * NaN is NaN: a right-hand side that broke down must not poison the work
  // vectors the next one starts from)
  void zero(T* y) { (void)hipMemsetAsync(y, 0, (size_t)n * sizeof(T), s); }
  H dot(const T* u, const T* w) {                  // <u, w> = u^H w
    hipLaunchKernelGGL(k_dot<T>, dim3(DOT_BLOCKS), dim3(256), 0, s, n, u, w, partial);
    T hp[DOT_BLOCKS];
    if (hipMemcpyAsync(hp, partial, sizeof(hp), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) { rc = PASTIX_AMD_ERR_DEVICE; return H(0.0); }
    H t(0.0);
    for (int b = 0; b < DOT_BLOCKS; b++) t += toH(hp[b]);
    return t;
  }
  double nrm(const T* u) { return std::sqrt(std::abs(dot(u, u))); }
  int precond(const T* r, T* z) {                  // z = (factors)^-1 r, through the factor's numbering
    hipLaunchKernelGGL(k_scatter_perm<T>, grid(), dim3(256), 0, s, n, perm, r, xp);
    const int rc2 = pastix_amd_solve_device(p, xp, 1);
    if (rc2) return rc2;
    hipLaunchKernelGGL(k_gather_perm<T>, grid(), dim3(256), 0, s, n, perm, xp, z);
    return 0;
  }
};

inline double conj_(double x) { return x; }
inline std::complex<double> conj_(const std::complex<double>& x) { return std::conj(x); }

template <class T>
int refine_impl(pastix_amd_plan_t* p, int mode, int sym, int64_t n, const int64_t* colptr, const int64_t* rows,
                const void* vals_, const int64_t* perm, const void* b_, void* x_, int64_t nrhs, double eps,
                int64_t itermax, int gmres_im, int64_t* iters_out, double* relerr_out) {
  using H = typename HostT<T>::type;
  const H* vals = (const H*)vals_;
  const bool herm = sym == 2;
  HIPCHK(hipSetDevice(p->device));
  Dev<T> D{p, p->stream, n};
  // ---- expand the CSC into a full CSR (row i lists a_ij; mirrored entries for symmetric / Hermitian input) ----
  {
    std::vector<int64_t> rp((size_t)n + 1, 0);
    for (int64_t j = 0; j < n; j++)
      for (int64_t q = colptr[j] - 1; q < colptr[j + 1] - 1; q++) {
        const int64_t i = rows[q] - 1;
        if (i < 0 || i >= n) return PASTIX_AMD_ERR_BADPARAMETER;
        rp[(size_t)i + 1]++;
        if (sym && i != j) rp[(size_t)j + 1]++;
      }
    for (int64_t i = 0; i < n; i++) rp[(size_t)i + 1] += rp[(size_t)i];
    const int64_t nz = rp[(size_t)n];
    std::vector<int32_t> ci((size_t)nz);
    std::vector<H> v((size_t)nz);
    std::vector<int64_t> pos(rp.begin(), rp.end() - 1);
    for (int64_t j = 0; j < n; j++)                       // columns ascending -> every row's entries come out sorted
      for (int64_t q = colptr[j] - 1; q < colptr[j + 1] - 1; q++) {
        const int64_t i = rows[q] - 1;
        ci[(size_t)pos[(size_t)i]] = (int32_t)j;
        v[(size_t)pos[(size_t)i]++] = vals[q];
        if (sym && i != j) {
          ci[(size_t)pos[(size_t)j]] = (int32_t)i;
          v[(size_t)pos[(size_t)j]++] = herm ? conj_(vals[q]) : vals[q];
        }
      }
    HIPCHK(hipMalloc((void**)&D.rp, ((size_t)n + 1) * sizeof(int64_t)));
    HIPCHK(hipMalloc((void**)&D.ci, (size_t)std::max<int64_t>(nz, 1) * sizeof(int32_t)));
    HIPCHK(hipMalloc((void**)&D.val, (size_t)std::max<int64_t>(nz, 1) * sizeof(T)));
    HIPCHK(hipMalloc((void**)&D.perm, (size_t)n * sizeof(int64_t)));
    HIPCHK(hipMalloc((void**)&D.partial, DOT_BLOCKS * sizeof(T)));
    HIPCHK(hipMemcpy(D.rp, rp.data(), ((size_t)n + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(D.ci, ci.data(), (size_t)nz * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(D.val, v.data(), (size_t)nz * sizeof(T), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(D.perm, perm, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice));
  }
  T *x = D.vec(), *f = D.vec(), *r = D.vec(), *z = D.vec(), *w = D.vec();
  D.xp = D.vec();
  if (D.rc) return D.rc;
  if (mode == 1 /* conjugate gradient */ && (p->host.factotype == PASTIX_AMD_FACT_LU || (!std::is_same<T, double>::value && !herm)))
    mode = 0;                                      // CG needs A = A^H
  // the mode's work vectors, ONCE for all right-hand sides (GMRES: m + 1 basis vectors of n entries -- 1.7 GB at n = 8e6
  // and the default m = 25 -- must not be allocated again per right-hand side beside the factor arenas)
  const int gm = (int)std::max<int64_t>(1, std::min<int64_t>(gmres_im > 0 ? gmres_im : 25, 200));
  T *pd = nullptr, *r0 = nullptr, *pv = nullptr, *v = nullptr, *sv = nullptr, *t = nullptr, *y = nullptr;
  std::vector<T*> V;
  if (mode == 1) {
    pd = D.vec();
  } else if (mode == 3) {
    r0 = D.vec(); pv = D.vec(); v = D.vec(); sv = D.vec(); t = D.vec(); y = D.vec();
  } else if (mode == 0) {
    V.resize((size_t)gm + 1);
    for (auto& q : V) q = D.vec();
  }
  if (D.rc) return D.rc;
  int64_t iters = 0;
  double relerr = 0;
  int rc = 0;
  for (int64_t c = 0; c < nrhs && !rc; c++) {
    HIPCHK(hipMemcpyAsync(x, (const H*)x_ + c * n, (size_t)n * sizeof(T), hipMemcpyHostToDevice, D.s));
    HIPCHK(hipMemcpyAsync(f, (const H*)b_ + c * n, (size_t)n * sizeof(T), hipMemcpyHostToDevice, D.s));
    double nb = D.nrm(f);
    if (nb == 0) nb = 1;
    auto residual = [&]() { D.spmv(x, r); D.axpby(H(1.0), f, H(-1.0), r); return D.nrm(r) / nb; };   // r = f - A x
    int64_t it = 0;
    relerr = residual();
    if (mode == 2) {                               // raff_pivot.c: x += M^-1 (b - A x)
      while (relerr >= eps && it < itermax) {
        if ((rc = D.precond(r, z))) break;
        D.axpby(H(1.0), z, H(1.0), x);
        it++;
        relerr = residual();
      }
    } else if (mode == 1) {                        // raff_grad.c: preconditioned conjugate gradient
      if ((rc = D.precond(r, z))) break;
      D.copy(z, pd);
      H rz = D.dot(r, z);
      while (relerr >= eps && it < itermax) {
        D.spmv(pd, w);
        const H alpha = rz / D.dot(pd, w);
        D.axpby(alpha, pd, H(1.0), x);
        D.axpby(-alpha, w, H(1.0), r);
        it++;
        relerr = D.nrm(r) / nb;
        if (relerr < eps) break;
        if ((rc = D.precond(r, z))) break;
        const H rz2 = D.dot(r, z);
        const H beta = rz2 / rz;
        rz = rz2;
        D.axpby(H(1.0), z, beta, pd);              // p = z + beta p
      }
      if (!rc) relerr = residual();
    } else if (mode == 3) {                        // raff_bicgstab.c: right-preconditioned BiCGStab
      D.copy(r, r0);
      H rho = 1.0, alpha = 1.0, omega = 1.0;
      D.zero(pv);                                  // p = 0
      D.zero(v);                                   // v = 0
      while (relerr >= eps && it < itermax) {
        const H rho1 = D.dot(r0, r);
        if (std::abs(rho1) == 0) break;            // breakdown: the true residual below decides
        const H beta = (rho1 / rho) * (alpha / omega);
        rho = rho1;
        D.axpby(-omega, v, H(1.0), pv);            // p = r + beta (p - omega v)
        D.axpby(H(1.0), r, beta, pv);
        if ((rc = D.precond(pv, y))) break;
        D.spmv(y, v);
        const H r0v = D.dot(r0, v);
        if (std::abs(r0v) == 0) break;
        alpha = rho / r0v;
        D.copy(r, sv);
        D.axpby(-alpha, v, H(1.0), sv);            // s = r - alpha v
        D.axpby(alpha, y, H(1.0), x);
        it++;
        relerr = D.nrm(sv) / nb;
        if (relerr < eps) break;
        if ((rc = D.precond(sv, z))) break;
        D.spmv(z, t);
        const H tt = D.dot(t, t);
        if (std::abs(tt) == 0) break;
        omega = D.dot(t, sv) / tt;
        D.axpby(omega, z, H(1.0), x);
        D.copy(sv, r);
        D.axpby(-omega, t, H(1.0), r);             // r = s - omega t
        relerr = D.nrm(r) / nb;
        if (std::abs(omega) == 0) break;
      }
      if (!rc) relerr = residual();
    } else {
      // raff_gmres.c: right-preconditioned GMRES(m): A M^-1 u = b, x = M^-1 u; modified Gram-Schmidt, Givens rotations
      const int m = gm;
      std::vector<H> Hm((size_t)(m + 1) * m), sn((size_t)m), g((size_t)m + 1), y((size_t)m);
      std::vector<double> cs((size_t)m);
      while (relerr >= eps && it < itermax && !rc) {
        const double beta = relerr * nb;
        D.axpby(H(1.0 / beta), r, H(0.0), V[0]);
        std::fill(g.begin(), g.end(), H(0.0));
        g[0] = beta;
        int j = 0;
        for (; j < m && it < itermax; j++) {
          if ((rc = D.precond(V[(size_t)j], z))) break;
          D.spmv(z, w);
          for (int i = 0; i <= j; i++) {
            const H h = D.dot(V[(size_t)i], w);
            Hm[(size_t)i * m + j] = h;
            D.axpby(-h, V[(size_t)i], H(1.0), w);
          }
          const double hn = D.nrm(w);
          Hm[(size_t)(j + 1) * m + j] = hn;
          if (hn > 0) D.axpby(H(1.0 / hn), w, H(0.0), V[(size_t)j + 1]);
          for (int i = 0; i < j; i++) {
            const H t = cs[(size_t)i] * Hm[(size_t)i * m + j] + sn[(size_t)i] * Hm[(size_t)(i + 1) * m + j];
            Hm[(size_t)(i + 1) * m + j] = -conj_(sn[(size_t)i]) * Hm[(size_t)i * m + j] + cs[(size_t)i] * Hm[(size_t)(i + 1) * m + j];
            Hm[(size_t)i * m + j] = t;
          }
          const H a0 = Hm[(size_t)j * m + j];
          const double a1 = hn, rr = std::sqrt(std::norm(std::complex<double>(a0)) + a1 * a1), a0abs = std::abs(a0);
          cs[(size_t)j] = rr > 0 ? a0abs / rr : 1.0;
          sn[(size_t)j] = rr > 0 ? (a0abs > 0 ? (a0 / a0abs) * (a1 / rr) : H(a1 / rr)) : H(0.0);
          Hm[(size_t)j * m + j] = cs[(size_t)j] * a0 + sn[(size_t)j] * a1;
          Hm[(size_t)(j + 1) * m + j] = 0.0;
          g[(size_t)j + 1] = -conj_(sn[(size_t)j]) * g[(size_t)j];
          g[(size_t)j] = cs[(size_t)j] * g[(size_t)j];
          it++;
          if (std::abs(g[(size_t)j + 1]) / nb < eps || hn == 0) { j++; break; }
        }
        if (rc) break;
        for (int i = j - 1; i >= 0; i--) {
          H t = g[(size_t)i];
          for (int q = i + 1; q < j; q++) t -= Hm[(size_t)i * m + q] * y[(size_t)q];
          y[(size_t)i] = t / Hm[(size_t)i * m + i];
        }
        D.zero(w);
        for (int i = 0; i < j; i++) D.axpby(y[(size_t)i], V[(size_t)i], H(1.0), w);
        if ((rc = D.precond(w, z))) break;
        D.axpby(H(1.0), z, H(1.0), x);
        relerr = residual();
      }
    }
    if (rc || D.rc) break;
    HIPCHK(hipMemcpyAsync((H*)x_ + c * n, x, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, D.s));
    HIPCHK(hipStreamSynchronize(D.s));
    iters = std::max(iters, it);
  }
  if (iters_out) *iters_out = iters;
  if (relerr_out) *relerr_out = relerr;
  HIPCHK(hipGetLastError());
  return rc ? rc : D.rc;
}

}  // namespace

extern "C" int pastix_amd_refine(pastix_amd_plan_t* p, int mode, int sym, pastix_amd_int_t n,
                                 const pastix_amd_int_t* colptr, const pastix_amd_int_t* rows, const void* vals,
                                 const pastix_amd_int_t* perm, const void* b, void* x, pastix_amd_int_t nrhs, double eps,
                                 pastix_amd_int_t itermax, int gmres_im, pastix_amd_int_t* iters, double* relerr) {
  if (!p || !colptr || !rows || !vals || !perm || !b || !x || n != p->host.ncol || nrhs < 1 || mode < 0 || mode > 3)
    return PASTIX_AMD_ERR_BADPARAMETER;
  if (p->distributed || p->host.opts.schur) return PASTIX_AMD_ERR_UNSUPPORTED;
  if (!p->factored) return PASTIX_AMD_ERR_BADPARAMETER;
  try {
    if (p->cplx) return refine_impl<zc>(p, mode, sym, n, colptr, rows, vals, perm, b, x, nrhs, eps, itermax, gmres_im, iters, relerr);
    return refine_impl<double>(p, mode, sym, n, colptr, rows, vals, perm, b, x, nrhs, eps, itermax, gmres_im, iters, relerr);
  } catch (const std::bad_alloc&) {
    return PASTIX_AMD_ERR_ALLOC;
  }
}
