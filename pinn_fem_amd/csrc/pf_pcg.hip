// pf_pcg.hip — matrix-free K(E,A) v and a Jacobi-preconditioned conjugate-gradient solve in float64:
// the linear solve inside the classical Newton-Raphson solver for scalar materials
// (FEM/python/fem/solver.py:408-512: `du_f = np.linalg.solve(k_ff, rhs_f)` on the dense float64 tangent
// of fem/assembly.py:16-75; fem/element.py:45-102).  SURVEY.md §8(f) rank 3: the reference's dense solve
// stops at ~2*10^4 dofs; here K is never formed.  The preconditioner is diag(K_ff) — the "Jacobi/diagonal
// preconditioner" BASELINE.json's north-star names (the reference itself has none).
//
// Layout: every vector is double [n_dofs]; fixed dofs are carried as zeros (K_ff is K with the fixed rows
// and columns dropped).  The element stiffness is ((double)E*(double)A)/(double)l0 with E, A from the
// per-element property arrays when a net is enabled, else the scalar value.  Sums over a node's elements
// run in ascending element id (no atomics: the same owner-computes gather as pf_mesh.hip).
#include "pf_common.h"

namespace {

enum { ST_RZ = 0, ST_PAP, ST_RR, ST_BB, ST_ALPHA, ST_BETA, ST_DONE, ST_ITERS, ST_RTOL2, ST_RZ_NEW, ST_COUNT = 16 };

__device__ __forceinline__ double elem_s64(const pf_problem& P, int e, float l0) {
  const double E = P.net[0].enabled ? (double)P.prop_e[e] : (double)P.net[0].scale;
  const double A = P.net[1].enabled ? (double)P.prop_a[e] : (double)P.net[1].scale;
  return (E * A) / (double)l0;
}

// (K v)[node] and diag(K)[node] in one pass over the node's elements
template <int DIM>
__device__ __forceinline__ void gather64(const pf_problem& P, const double* __restrict__ v, int node,
                                         double* kv, double* diag) {
  const pf_mesh& M = P.mesh;
#pragma unroll
  for (int c = 0; c < DIM; ++c) { kv[c] = 0.0; diag[c] = 0.0; }
  for (int idx = M.adj_ptr[node]; idx < M.adj_ptr[node + 1]; ++idx) {
    const int code = M.adj[idx];
    const int e = code >> 1, end = code & 1;
    const int2 nn = reinterpret_cast<const int2*>(M.conn)[e];
    const ElemGeo g = load_geo(M.egeo, e);
    const double s = elem_s64(P, e, g.l0);
    const double sg = end ? -1.0 : 1.0;
    if (DIM == 2) {
      const double dx = v ? v[2 * nn.y] - v[2 * nn.x] : 0.0, dy = v ? v[2 * nn.y + 1] - v[2 * nn.x + 1] : 0.0;
      // rows of s*pattern @ [v_i; v_j] for this end: -(sg*s) * (c2*dx + cs*dy), -(sg*s) * (cs*dx + s2*dy)
      kv[0] += -(sg * s) * ((double)g.c2 * dx + (double)g.cs * dy);
      kv[1] += -(sg * s) * ((double)g.cs * dx + (double)g.s2 * dy);
      diag[0] += s * (double)g.c2;
      diag[1] += s * (double)g.s2;
    } else {
      const double dx = v ? v[nn.y] - v[nn.x] : 0.0;
      kv[0] += -(sg * s) * dx;
      diag[0] += s;
    }
  }
}

template <int DIM>
__global__ __launch_bounds__(256) void k_kv64(pf_problem P, const double* __restrict__ v, double* __restrict__ out,
                                              int zero_fixed) {
  const pf_mesh& M = P.mesh;
  for (int node = blockIdx.x * blockDim.x + threadIdx.x; node < M.n_nodes; node += gridDim.x * blockDim.x) {
    double kv[2], dg[2];
    gather64<DIM>(P, v, node, kv, dg);
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      const int dof = node * DIM + c;
      out[dof] = (zero_fixed && (M.dof_flags[dof] & PF_DOF_FIXED)) ? 0.0 : kv[c];
    }
  }
}

__device__ __forceinline__ double block_sum64(double v, double* smem) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if (lane == 0) smem[w] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < nw; ++i) t += smem[i];
  return t;
}

// x = 0, r = b (free dofs), dinv = 1/diag(K_ff), z = dinv*r, p = z; partials of r.z and b.b
template <int DIM>
__global__ __launch_bounds__(256) void k_pcg_init(pf_problem P, const double* __restrict__ b, double* x, double* r,
                                                  double* z, double* p, double* dinv, double* part) {
  __shared__ double red[8];
  const pf_mesh& M = P.mesh;
  double rz = 0.0, bb = 0.0;
  for (int node = blockIdx.x * blockDim.x + threadIdx.x; node < M.n_nodes; node += gridDim.x * blockDim.x) {
    double kv[2], dg[2];
    gather64<DIM>(P, nullptr, node, kv, dg);
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      const int dof = node * DIM + c;
      const bool fixed = M.dof_flags[dof] & PF_DOF_FIXED;
      const double bi = fixed ? 0.0 : b[dof];
      const double di = (fixed || dg[c] == 0.0) ? 0.0 : 1.0 / dg[c];
      x[dof] = 0.0; r[dof] = bi; dinv[dof] = di;
      const double zi = di * bi;
      z[dof] = zi; p[dof] = zi;
      rz += bi * zi; bb += bi * bi;
    }
  }
  const double t0 = block_sum64(rz, red), t1 = block_sum64(bb, red);
  if (threadIdx.x == 0) { part[blockIdx.x] = t0; part[PF_NODE_SLOTS + blockIdx.x] = t1; }
}

// ap = K p (fixed rows zero); partial p.ap
template <int DIM>
__global__ __launch_bounds__(256) void k_pcg_ap(pf_problem P, const double* __restrict__ st, const double* __restrict__ p,
                                                double* __restrict__ ap, double* part) {
  if (st[ST_DONE] != 0.0) return;
  __shared__ double red[8];
  const pf_mesh& M = P.mesh;
  double pap = 0.0;
  for (int node = blockIdx.x * blockDim.x + threadIdx.x; node < M.n_nodes; node += gridDim.x * blockDim.x) {
    double kv[2], dg[2];
    gather64<DIM>(P, p, node, kv, dg);
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      const int dof = node * DIM + c;
      const double a = (M.dof_flags[dof] & PF_DOF_FIXED) ? 0.0 : kv[c];
      ap[dof] = a;
      pap += p[dof] * a;
    }
  }
  const double t = block_sum64(pap, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// one block: phase 0 (after init) rz, bb | phase 1 (after ap) pAp -> alpha | phase 2 (after update) rz_new, rr ->
// beta, stop test
__global__ __launch_bounds__(1024) void k_pcg_scalars(double* st, const double* __restrict__ part, int nb, int phase) {
  if (phase != 0 && st[ST_DONE] != 0.0) return;
  __shared__ double red[16];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) { a += part[i]; b += part[PF_NODE_SLOTS + i]; }
  const double ta = block_sum64(a, red), tb = block_sum64(b, red);
  if (threadIdx.x != 0) return;
  if (phase == 0) {
    st[ST_RZ] = ta; st[ST_BB] = tb; st[ST_RR] = tb; st[ST_ITERS] = 0.0;
    st[ST_DONE] = (tb == 0.0) ? 1.0 : 0.0;        // b = 0: x = 0 is the solution
  } else if (phase == 1) {
    st[ST_PAP] = ta;
    st[ST_ALPHA] = ta != 0.0 ? st[ST_RZ] / ta : 0.0;
  } else {
    st[ST_BETA] = st[ST_RZ] != 0.0 ? ta / st[ST_RZ] : 0.0;
    st[ST_RZ] = ta; st[ST_RR] = tb; st[ST_ITERS] += 1.0;
    if (tb <= st[ST_RTOL2] * st[ST_BB] || ta == 0.0) st[ST_DONE] = 1.0;
  }
}

// x += alpha p; r -= alpha ap; z = dinv r; partials r.z, r.r
__global__ __launch_bounds__(256) void k_pcg_update(const double* __restrict__ st, int n, double* x, double* r, double* z,
                                                    const double* __restrict__ p, const double* __restrict__ ap,
                                                    const double* __restrict__ dinv, double* part) {
  if (st[ST_DONE] != 0.0) return;
  __shared__ double red[8];
  const double alpha = st[ST_ALPHA];
  double rz = 0.0, rr = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * ap[i];
    r[i] = ri;
    const double zi = dinv[i] * ri;
    z[i] = zi;
    rz += ri * zi; rr += ri * ri;
  }
  const double t0 = block_sum64(rz, red), t1 = block_sum64(rr, red);
  if (threadIdx.x == 0) { part[blockIdx.x] = t0; part[PF_NODE_SLOTS + blockIdx.x] = t1; }
}

// p = z + beta p
__global__ __launch_bounds__(256) void k_pcg_dir(const double* __restrict__ st, int n, const double* __restrict__ z, double* p) {
  if (st[ST_DONE] != 0.0) return;
  const double beta = st[ST_BETA];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = z[i] + beta * p[i];
}

}  // namespace

#define PCG_CHECK(what)                                                      \
  if (hipGetLastError() != hipSuccess) { pf_set_error(what ": HIP launch failed"); return PF_ERR_HIP; }

extern "C" {

long long pf_pcg_workspace_count(const pf_problem* p) {
  if (!p) return PF_ERR_ARG;
  return 5LL * p->mesh.n_dofs + 2LL * PF_NODE_SLOTS + ST_COUNT;   // r, z, p, ap, dinv | partials | state
}

int pf_kv_f64(const pf_problem* p, const double* v, double* out, int zero_fixed, void* stream) {
  if (!p || !v || !out) { pf_set_error("pf_kv_f64: null argument"); return PF_ERR_ARG; }
  const int nb = pf_node_blocks(p->mesh.n_nodes);
  hipStream_t s = (hipStream_t)stream;
  if (p->mesh.dim == 2) hipLaunchKernelGGL(k_kv64<2>, dim3(nb), dim3(256), 0, s, *p, v, out, zero_fixed);
  else hipLaunchKernelGGL(k_kv64<1>, dim3(nb), dim3(256), 0, s, *p, v, out, zero_fixed);
  PCG_CHECK("pf_kv_f64");
  return PF_OK;
}

// ws layout (doubles): r | z | p | ap | dinv (n_dofs each) | partials (2*PF_NODE_SLOTS) | state (16)
int pf_pcg_begin(const pf_problem* p, const double* b, double* x, double* ws, double rtol, void* stream) {
  if (!p || !b || !x || !ws || !(rtol >= 0.0)) { pf_set_error("pf_pcg_begin: bad argument"); return PF_ERR_ARG; }
  const int n = p->mesh.n_dofs, nb = pf_node_blocks(p->mesh.n_nodes);
  double *r = ws, *z = ws + n, *pp = ws + 2 * (size_t)n, *dinv = ws + 4 * (size_t)n;
  double* part = ws + 5 * (size_t)n;
  double* st = part + 2 * PF_NODE_SLOTS;
  hipStream_t s = (hipStream_t)stream;
  const double rtol2 = rtol * rtol;
  if (hipMemsetAsync(st, 0, ST_COUNT * sizeof(double), s) != hipSuccess ||
      hipMemcpyAsync(st + ST_RTOL2, &rtol2, sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess) {
    pf_set_error("pf_pcg_begin: state setup failed");
    return PF_ERR_HIP;
  }
  if (hipStreamSynchronize(s) != hipSuccess) { pf_set_error("pf_pcg_begin: sync failed"); return PF_ERR_HIP; }  // rtol2 lives on the stack
  if (p->mesh.dim == 2) hipLaunchKernelGGL(k_pcg_init<2>, dim3(nb), dim3(256), 0, s, *p, b, x, r, z, pp, dinv, part);
  else hipLaunchKernelGGL(k_pcg_init<1>, dim3(nb), dim3(256), 0, s, *p, b, x, r, z, pp, dinv, part);
  PCG_CHECK("pf_pcg_begin");
  hipLaunchKernelGGL(k_pcg_scalars, dim3(1), dim3(1024), 0, s, st, part, nb, 0);
  PCG_CHECK("pf_pcg_begin");
  return PF_OK;
}

static int pcg_enqueue(const pf_problem* p, double* x, double* ws, int n_iter, hipStream_t s) {
  const int n = p->mesh.n_dofs, nb = pf_node_blocks(p->mesh.n_nodes);
  double *r = ws, *z = ws + n, *pp = ws + 2 * (size_t)n, *ap = ws + 3 * (size_t)n, *dinv = ws + 4 * (size_t)n;
  double* part = ws + 5 * (size_t)n;
  double* st = part + 2 * PF_NODE_SLOTS;
  int nbv = (n + 255) / 256;
  if (nbv > PF_MAX_NODE_BLOCKS) nbv = PF_MAX_NODE_BLOCKS;
  for (int it = 0; it < n_iter; ++it) {
    if (p->mesh.dim == 2) hipLaunchKernelGGL(k_pcg_ap<2>, dim3(nb), dim3(256), 0, s, *p, st, pp, ap, part);
    else hipLaunchKernelGGL(k_pcg_ap<1>, dim3(nb), dim3(256), 0, s, *p, st, pp, ap, part);
    hipLaunchKernelGGL(k_pcg_scalars, dim3(1), dim3(1024), 0, s, st, part, nb, 1);
    hipLaunchKernelGGL(k_pcg_update, dim3(nbv), dim3(256), 0, s, st, n, x, r, z, pp, ap, dinv, part);
    hipLaunchKernelGGL(k_pcg_scalars, dim3(1), dim3(1024), 0, s, st, part, nbv, 2);
    hipLaunchKernelGGL(k_pcg_dir, dim3(nbv), dim3(256), 0, s, st, n, z, pp);
  }
  PCG_CHECK("pf_pcg_iterations");
  return PF_OK;
}

static int pcg_read_state(double* ws, int n, double* state_out, hipStream_t s) {
  double* st = ws + 5 * (size_t)n + 2 * PF_NODE_SLOTS;
  double h[ST_COUNT];
  if (hipMemcpyAsync(h, st, sizeof(h), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
    pf_set_error("pf_pcg: state read-back failed");
    return PF_ERR_HIP;
  }
  state_out[0] = h[ST_ITERS]; state_out[1] = h[ST_DONE]; state_out[2] = h[ST_RR]; state_out[3] = h[ST_BB];
  return PF_OK;
}

// n_iter CG iterations (no-ops once the stop test |r| <= rtol |b| fired); state_out (host, may be NULL)
// receives [iterations, done, |r|^2, |b|^2] after a stream synchronisation
int pf_pcg_iterations(const pf_problem* p, double* x, double* ws, int n_iter, double* state_out, void* stream) {
  if (!p || !x || !ws || n_iter < 0) { pf_set_error("pf_pcg_iterations: bad argument"); return PF_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  int rc = pcg_enqueue(p, x, ws, n_iter, s);
  if (rc != PF_OK) return rc;
  return state_out ? pcg_read_state(ws, p->mesh.n_dofs, state_out, s) : PF_OK;
}

// the same n_iter iterations as ONE hipGraph (record and pointers baked in; handle for pf_graph_launch /
// pf_graph_destroy): 5 tiny launches per CG iteration are launch bound when issued one by one
int pf_pcg_graph_create(const pf_problem* p, double* x, double* ws, int n_iter, void* stream, void** graph_out) {
  if (!p || !x || !ws || n_iter < 1 || !graph_out) { pf_set_error("pf_pcg_graph_create: bad argument"); return PF_ERR_ARG; }
  hipStream_t s = (hipStream_t)stream;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    pf_set_error("pf_pcg_graph_create: hipStreamBeginCapture failed");
    return PF_ERR_HIP;
  }
  const int rc = pcg_enqueue(p, x, ws, n_iter, s);
  const hipError_t e = hipStreamEndCapture(s, &graph);
  if (rc != PF_OK || e != hipSuccess || !graph) {
    if (graph) hipGraphDestroy(graph);
    if (rc == PF_OK) pf_set_error("pf_pcg_graph_create: capture failed");
    return rc != PF_OK ? rc : PF_ERR_HIP;
  }
  if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
    hipGraphDestroy(graph);
    pf_set_error("pf_pcg_graph_create: hipGraphInstantiate failed");
    return PF_ERR_HIP;
  }
  hipGraphDestroy(graph);
  *graph_out = (void*)exec;
  return PF_OK;
}

// [iterations, stopped, |r|^2, |b|^2] of the running solve (synchronises the stream)
int pf_pcg_state(const pf_problem* p, double* ws, double* state_out, void* stream) {
  if (!p || !ws || !state_out) { pf_set_error("pf_pcg_state: bad argument"); return PF_ERR_ARG; }
  return pcg_read_state(ws, p->mesh.n_dofs, state_out, (hipStream_t)stream);
}

}  // extern "C"
