// pf_api.hip — the C ABI declared in include/pinnfem_hip.h: argument checks, net-shape dispatch,
// and the fused GD-iteration launch sequence (FEM/python/fem/solver.py:254-355).
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "pf_common.h"
#include "pf_net32.h"

// launchers from pf_mesh.hip
int pf_launch_node_residual(const pf_problem* p, float* f_int_out, int compute_loss, hipStream_t s, int fin_prev = 0);
int pf_launch_elem_adjoint(const pf_problem* p, hipStream_t s);
int pf_launch_node_gradu(const pf_problem* p, int fuse_adam, hipStream_t s, int skip_shared = 0, float* u_out = nullptr);
int pf_launch_u_home(const pf_problem* p, hipStream_t s);
int pf_launch_scalar_residual(const pf_problem* p, const pf_scalar_id* sp, hipStream_t s);
int pf_launch_scalar_update(const pf_problem* p, const pf_scalar_id* sp, hipStream_t s);
int pf_launch_shard_pack(const pf_problem* p, float* buf2, const float* u2_local, hipStream_t s);
int pf_launch_shard_update(const pf_problem* p, const float* buf2, float* u2_local, hipStream_t s);
int pf_launch_shard_flush(const pf_problem* p, const float* u2, hipStream_t s);
int pf_launch_theta_reduce(const pf_problem* p, int fuse_adam, hipStream_t s);
int pf_launch_pack_theta(const pf_problem* p, hipStream_t s);
int pf_launch_finalize(const pf_problem* p, int mode, int with_theta, hipStream_t s, int tn_ready = 0);
int pf_launch_theta_stage1(const pf_problem* p, hipStream_t s);
int pf_launch_theta_stage2(const pf_problem* p, int fuse_adam, hipStream_t s);
int pf_launch_reset(const pf_problem* p, hipStream_t s);
int pf_launch_adam(float* param, const float* grad, float* m, float* v, int n, int step, double lr,
                   double beta1, double beta2, double eps, hipStream_t s);
int pf_launch_diag_k(const pf_problem* p, float* diag, hipStream_t s);
int pf_launch_dense_k(const pf_problem* p, float* K, hipStream_t s);
int pf_launch_coo_k(const pf_problem* p, long long* rows, long long* cols, float* vals, hipStream_t s);

// below this many elements the iteration graph is a plain chain: the kernels are then too short to hide anything behind,
// the branches' fork/join cost (~5 us each against ~1.5 us for a plain boundary) would only add to a launch-bound iteration
#define PF_GRAPH_DAG_MIN_ELEMS 200000   /* measured crossover between 1e5 (chain 0.061 vs DAG 0.067 ms) and 3e5 (0.099 vs 0.092) */

static thread_local char g_err[512] = "";

void pf_set_error(const char* msg) {
  strncpy(g_err, msg, sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

static int fail(int code, const char* msg) {
  pf_set_error(msg);
  return code;
}

static int check_launch(int rc, const char* what) {
  if (rc == PF_OK) return PF_OK;
  if (rc == PF_ERR_HIP) {
    char buf[400];
    snprintf(buf, sizeof(buf), "%s: HIP launch failed: %s", what, hipGetErrorString(hipGetLastError()));
    pf_set_error(buf);
  }
  return rc;
}

static int padded_width(int width) {
  if (width < 1 || width > 32) return PF_ERR_UNSUPPORTED;
  return ((width + 3) / 4) * 4;
}

static int check_problem(const pf_problem* p) {
  if (!p) return fail(PF_ERR_ARG, "null problem");
  const pf_mesh& M = p->mesh;
  if (M.dim != 1 && M.dim != 2) return fail(PF_ERR_ARG, "mesh.dim must be 1 or 2");
  if (M.n_nodes < 1 || M.n_elems < 0 || M.n_dofs != M.n_nodes * M.dim)
    return fail(PF_ERR_ARG, "inconsistent mesh sizes");
  if (!M.conn || !M.egeo || !M.ecent || !M.adj_ptr || !M.adj || !M.f_ext || !M.dof_flags || !M.meas_val)
    return fail(PF_ERR_ARG, "null mesh array");
  if (p->n_part_blocks < 1 || p->n_part_blocks > PF_MAX_BLOCKS)
    return fail(PF_ERR_ARG, "n_part_blocks out of range");
  if (!p->u || !p->g_f || !p->partials || !p->state) return fail(PF_ERR_ARG, "null state/workspace");
  for (int k = 0; k < 2; ++k) {
    const pf_net& n = p->net[k];
    if (!n.enabled) continue;
    if (n.in_dim != M.dim + 1)
      return fail(PF_ERR_ARG, "net in_dim must be mesh.dim+1 (columns load_factor, x[, y])");
    if (padded_width(n.width) < 0 || n.n_hidden < 1 || n.n_hidden > 3)
      return fail(PF_ERR_UNSUPPORTED, "net shape outside the compiled menu (width 1..32, hidden layers 1..3)");
    if (!p->theta || !p->theta_pad || !p->pad_index || !p->g_ea || !p->grad_theta)
      return fail(PF_ERR_ARG, "null parameter workspace");
    if ((k == 0 && !p->prop_e) || (k == 1 && !p->prop_a)) return fail(PF_ERR_ARG, "null property array");
    if (p->wg_mode == PF_WG_MFMA32) {
      if (n.width > PF_N32_WIDTH_MAX)
        return fail(PF_ERR_UNSUPPORTED, "MFMA32 engine supports widths up to 30 (use PF_WG_MFMA44 beyond)");
      if (!p->net_op) return fail(PF_ERR_ARG, "MFMA32 engine needs the operand image workspace (net_op)");
      if (p->mlp_dtype != PF_MLP_F32 && p->mlp_dtype != PF_MLP_BF16) return fail(PF_ERR_ARG, "mlp_dtype must be PF_MLP_F32 or PF_MLP_BF16");
    }
  }
  return PF_OK;
}

#define PF_WIDTH_SWITCH(PREFIX)                                   \
  switch (padded_width(p->net[which].width)) {                    \
    case 4: return PREFIX##4(p, which, s);                        \
    case 8: return PREFIX##8(p, which, s);                        \
    case 12: return PREFIX##12(p, which, s);                      \
    case 16: return PREFIX##16(p, which, s);                      \
    case 20: return PREFIX##20(p, which, s);                      \
    case 24: return PREFIX##24(p, which, s);                      \
    case 28: return PREFIX##28(p, which, s);                      \
    case 32: return PREFIX##32(p, which, s);                      \
  }                                                               \
  return fail(PF_ERR_UNSUPPORTED, "net width outside 1..32");

// MFMA32 engine: one translation unit per register bucket
#define PF_NR_SWITCH(PREFIX)                                      \
  switch (pf_net32_bucket(p->net[which].width)) {                 \
    case 2: return PREFIX##2(p, which, s);                        \
    case 4: return PREFIX##4(p, which, s);                        \
    case 6: return PREFIX##6(p, which, s);                        \
    case 8: return PREFIX##8(p, which, s);                        \
    case 10: return PREFIX##10(p, which, s);                      \
    case 12: return PREFIX##12(p, which, s);                      \
    case 15: return PREFIX##15(p, which, s);                      \
  }                                                               \
  return fail(PF_ERR_UNSUPPORTED, "MFMA32 engine: net width outside 1..30");

static int net_forward_impl(const pf_problem* p, int which, hipStream_t s, int s2_half);
// write_s == 0: this forward is followed by the other net's, which writes the stiffness records (pf_problem.elem_k).
// s2_half >= 0 (MFMA32 engine): the launch first runs the previous iteration's parameter update from that state half.
static int net_forward(const pf_problem* p, int which, hipStream_t s, int write_s = 1, int s2_half = -1) {
  if (write_s || !p->elem_k) return net_forward_impl(p, which, s, s2_half);
  pf_problem q = *p;
  q.elem_k = nullptr;
  return net_forward_impl(&q, which, s, s2_half);
}
#define PF_NR_SWITCH_FWD(PREFIX)                                  \
  switch (pf_net32_bucket(p->net[which].width)) {                 \
    case 2: return PREFIX##2(p, which, s, s2_half);               \
    case 4: return PREFIX##4(p, which, s, s2_half);               \
    case 6: return PREFIX##6(p, which, s, s2_half);               \
    case 8: return PREFIX##8(p, which, s, s2_half);               \
    case 10: return PREFIX##10(p, which, s, s2_half);             \
    case 12: return PREFIX##12(p, which, s, s2_half);             \
    case 15: return PREFIX##15(p, which, s, s2_half);             \
  }                                                               \
  return fail(PF_ERR_UNSUPPORTED, "MFMA32 engine: net width outside 1..30");
static int net_forward_impl(const pf_problem* p, int which, hipStream_t s, int s2_half) {
  if (p->wg_mode == PF_WG_MFMA32 && p->mlp_dtype == PF_MLP_BF16) { PF_NR_SWITCH_FWD(pf_launch_net32b_forward_) }
  if (p->wg_mode == PF_WG_MFMA32) { PF_NR_SWITCH_FWD(pf_launch_net32_forward_) }
  if (s2_half >= 0) return fail(PF_ERR_ARG, "the fused parameter update exists in the MFMA32 engine only");
  if (p->wg_mode == PF_WG_MFMA44) { PF_WIDTH_SWITCH(pf_launch_net44_forward_) }
  PF_WIDTH_SWITCH(pf_launch_net_forward_)
}

// both nets in ONE launch (pf_net32.hip: k_net32_forward2) where the engine has it: MFMA32, both nets enabled, same
// number of hidden layers and inputs.  PF_FUSE_FWD=0: experiment knob (two launches, as before round 3).
static bool can_fuse_forward(const pf_problem* p) {
  static const int knob = getenv("PF_FUSE_FWD") ? atoi(getenv("PF_FUSE_FWD")) : 1;
  return knob != 0 && p->wg_mode == PF_WG_MFMA32 && p->net[0].enabled && p->net[1].enabled &&
         p->net[0].n_hidden == p->net[1].n_hidden && p->net[0].in_dim == p->net[1].in_dim;
}
#define PF_NR0_SWITCH(PREFIX)                                     \
  switch (pf_net32_bucket(p->net[0].width)) {                     \
    case 2: return PREFIX##2(p, s, o);  \
    case 4: return PREFIX##4(p, s, o);  \
    case 6: return PREFIX##6(p, s, o);  \
    case 8: return PREFIX##8(p, s, o);  \
    case 10: return PREFIX##10(p, s, o);  \
    case 12: return PREFIX##12(p, s, o);  \
    case 15: return PREFIX##15(p, s, o);  \
  }                                                               \
  return fail(PF_ERR_UNSUPPORTED, "MFMA32 engine: net width outside 1..30");
static int net_forward2(const pf_problem* p, hipStream_t s, const pf_fwd2_opts& o = pf_fwd2_opts()) {
  if (p->mlp_dtype == PF_MLP_BF16) { PF_NR0_SWITCH(pf_launch_net32b_forward2_) }
  PF_NR0_SWITCH(pf_launch_net32_forward2_)
}
// the forward pass of every enabled net: the properties and, with the MFMA32 engine, the stiffness records.
// s2_half >= 0: the FIRST launch also runs the parameter update of the previous iteration (can_fuse_theta_update)
// o.gu_nb (only with can_fuse_gradu; fused launch only): see pf_fwd2_opts
static int net_forward_all(const pf_problem* p, hipStream_t s, const pf_fwd2_opts& o = pf_fwd2_opts()) {
  if (can_fuse_forward(p)) return net_forward2(p, s, o);
  int s2_half = o.s2_half >= 0 && o.calc_index ? o.s2_half + 2 : o.s2_half;      // (+ 2: see k_net32_forward)
  for (int k = 0; k < 2; ++k)
    if (p->net[k].enabled) {
      const int rc = net_forward(p, k, s, k == 1 || !p->net[1].enabled, s2_half);
      if (rc != PF_OK) return rc;
      s2_half = -1;
    }
  return PF_OK;
}
// Can the iteration graph fold the parameter update of iteration t into the forward launch of t+1?  MFMA32 engine with
// the second state half present; PF_FUSE_S2=0: experiment knob (stand-alone update launch every iteration).
static bool can_fuse_theta_update(const pf_problem* p) {
  static const int knob = getenv("PF_FUSE_S2") ? atoi(getenv("PF_FUSE_S2")) : 1;
  return knob != 0 && p->wg_mode == PF_WG_MFMA32 && p->theta_alt != nullptr && p->n_theta_active > 0 &&
         (p->net[0].enabled || p->net[1].enabled);
}

// Can the iteration graph fold the displacement update (dL/du + Adam(u) + clamp) of iteration t into the fused forward
// launch of t+1 (pf_net32.hip: k_net32_forward2, gu_nb; pf_node.h)?  The graph is then a plain chain: no side branch, no
// fork, no join, no second displacement vector.  PF_FUSE_GU=0: experiment knob (gradu on its own branch, as in round 2).
static bool can_fuse_gradu(const pf_problem* p) {
  static const int knob = getenv("PF_FUSE_GU") ? atoi(getenv("PF_FUSE_GU")) : 1;
  return knob != 0 && can_fuse_forward(p) && pf_n32_fwd2_can_update_u(p, pf_node_blocks(p->mesh.n_nodes));
}

static int net_backward(const pf_problem* p, int which, hipStream_t s) {
  if (p->wg_mode == PF_WG_MFMA32 && p->mlp_dtype == PF_MLP_BF16) { PF_NR_SWITCH(pf_launch_net32b_backward_) }
  if (p->wg_mode == PF_WG_MFMA32) { PF_NR_SWITCH(pf_launch_net32_backward_) }
  if (p->wg_mode == PF_WG_MFMA44) { PF_WIDTH_SWITCH(pf_launch_net44_backward_) }
  PF_WIDTH_SWITCH(pf_launch_net_backward_)
}

// Is the element adjoint (dL/d(EA) per element) computed inside the first net's backward kernel?
static bool fuse_gea_for(const pf_problem* p) {
  static const int knob = getenv("PF_FUSE_GEA") ? atoi(getenv("PF_FUSE_GEA")) : -1;   // experiment knob
  if (p->wg_mode == PF_WG_MFMA44) return true;
  if (p->wg_mode == PF_WG_MFMA32) return knob < 0 ? true : knob != 0;
  return false;
}
// backward that also computes and stores the element adjoint g_ea
static int net_backward_gea(const pf_problem* p, int which, hipStream_t s) {
  if (p->wg_mode == PF_WG_MFMA32 && p->mlp_dtype == PF_MLP_BF16) { PF_NR_SWITCH(pf_launch_net32b_backward_gea_) }
  if (p->wg_mode == PF_WG_MFMA32) { PF_NR_SWITCH(pf_launch_net32_backward_gea_) }
  PF_WIDTH_SWITCH(pf_launch_net44_backward_gea_)
}

// both backward passes (the first with the fused element adjoint) in ONE launch, two phases (pf_net32.hip:
// k_net32_backward2).  PF_FUSE_BWD=0: experiment knob.
static bool can_fuse_backward(const pf_problem* p) {
  static const int knob = getenv("PF_FUSE_BWD") ? atoi(getenv("PF_FUSE_BWD")) : 1;
  return knob != 0 && can_fuse_forward(p) && p->net[0].n_hidden == 2 && fuse_gea_for(p);
}
// reduce_rows != 0: the launch also does theta stage 1 (the last block of every row group sums the group's rows into the
// second-level row, the arithmetic of k_theta_stage1): no pf_launch_theta_stage1 behind it.  OFF by default
// (PF_FUSE_S1=1 switches it on): bit-identical, but measured SLOWER on MI355X — the backward launch grew from 79 to 92 us
// (every block drains its stores, takes a ticket, and the 16 last blocks read 16 rows each through sc1 loads) for a 7 us
// kernel saved: 0.1547 against 0.1500 ms per iteration (profiles/r03_ab.txt).
static bool fuse_s1_knob() {
  static const int knob = getenv("PF_FUSE_S1") ? atoi(getenv("PF_FUSE_S1")) : 0;
  return knob != 0;
}
static int net_backward2(const pf_problem* p, hipStream_t s, int reduce_rows = 0) {
#define PF_NR0_SWITCH_B(PREFIX)                                   \
  switch (pf_net32_bucket(p->net[0].width)) {                     \
    case 2: return PREFIX##2(p, s, reduce_rows);                  \
    case 4: return PREFIX##4(p, s, reduce_rows);                  \
    case 6: return PREFIX##6(p, s, reduce_rows);                  \
    case 8: return PREFIX##8(p, s, reduce_rows);                  \
    case 10: return PREFIX##10(p, s, reduce_rows);                \
    case 12: return PREFIX##12(p, s, reduce_rows);                \
    case 15: return PREFIX##15(p, s, reduce_rows);                \
  }                                                               \
  return fail(PF_ERR_UNSUPPORTED, "MFMA32 engine: net width outside 1..30");
  if (p->mlp_dtype == PF_MLP_BF16) { PF_NR0_SWITCH_B(pf_launch_net32b_backward2_) }
  PF_NR0_SWITCH_B(pf_launch_net32_backward2_)
}
// element adjoint + backward of every enabled net; with the MFMA44 engine the adjoint is fused into
// the first net's backward kernel


#define PF_TRY(expr, what)                       \
  do {                                           \
    int rc__ = check_launch((expr), what);       \
    if (rc__ != PF_OK) return rc__;              \
  } while (0)

extern "C" {

int pf_abi_version(void) { return PF_ABI_VERSION; }
const char* pf_last_error(void) { return g_err; }

int pf_sizeof(int what) {
  switch (what) {
    case 0: return (int)sizeof(pf_mesh);
    case 1: return (int)sizeof(pf_net);
    case 2: return (int)sizeof(pf_state);
    case 3: return (int)sizeof(pf_problem);
    case 4: return (int)sizeof(pf_scalar_id);
  }
  return PF_ERR_ARG;
}

int pf_net_param_count(int in_dim, int width, int n_hidden) {
  if (in_dim < 1 || width < 1 || n_hidden < 1) return PF_ERR_ARG;
  return width * in_dim + width + (n_hidden - 1) * (width * width + width) + width + 1;
}

int pf_padded_width(int width) { return padded_width(width); }

int pf_net_pad_count(int in_dim, int width, int n_hidden) {
  const int hp = padded_width(width);
  if (hp < 0 || n_hidden < 1 || n_hidden > 3 || (in_dim != 2 && in_dim != 3)) return PF_ERR_UNSUPPORTED;
  return pf_pad_count(hp, n_hidden);
}

// padded-image index of torch parameter `local` (0-based inside the net's parameters() order)
int pf_net_pad_index(int in_dim, int width, int n_hidden, int local) {
  const int hp = padded_width(width);
  if (hp < 0 || n_hidden < 1 || n_hidden > 3 || (in_dim != 2 && in_dim != 3)) return PF_ERR_UNSUPPORTED;
  int q = local;
  if (q < 0) return PF_ERR_ARG;
  // W1 (width,in_dim), b1 (width)
  if (q < width * in_dim) return (q / in_dim) * 4 + (q % in_dim);
  q -= width * in_dim;
  if (q < width) return q * 4 + in_dim;
  q -= width;
  for (int l = 2; l <= n_hidden; ++l) {
    if (q < width * width) return pf_pad_wh(hp, l) + (q / width) * (hp + 4) + (q % width);
    q -= width * width;
    if (q < width) return pf_pad_wh(hp, l) + q * (hp + 4) + hp;
    q -= width;
  }
  if (q < width) return pf_pad_wo(hp, n_hidden) + q;
  q -= width;
  if (q == 0) return pf_pad_wo(hp, n_hidden) + hp;
  return PF_ERR_ARG;
}

int pf_net_op_count(int in_dim, int width, int n_hidden) {
  if (width < 1 || width > PF_N32_WIDTH_MAX || n_hidden < 1 || n_hidden > 3 || (in_dim != 2 && in_dim != 3))
    return PF_ERR_UNSUPPORTED;
  return (pf_n32_bytes(n_hidden) + 3) / 4;
}

long long pf_partials_count(const pf_problem* p) {
  if (!p) return PF_ERR_ARG;
  return (long long)PF_PART_WG + ((long long)p->n_part_blocks + PF_RG) * (long long)p->pad_total + PF_TICKETS;
}

int pf_fusion_info(const pf_problem* p) {
  if (check_problem(p) != PF_OK) return PF_ERR_ARG;
  int m = 0;
  if (can_fuse_forward(p)) m |= PF_FUSED_FORWARD;
  if (can_fuse_backward(p)) m |= PF_FUSED_BACKWARD;
  if (can_fuse_theta_update(p)) m |= PF_FUSED_THETA_UPDATE;
  if (p->prop_double != 0 && p->elem_k != nullptr && can_fuse_gradu(p)) m |= PF_FUSED_U_UPDATE;
  else if (p->u_alt != nullptr && p->mesh.n_elems >= PF_GRAPH_DAG_MIN_ELEMS) m |= PF_FUSED_U_PINGPONG;
  return m;
}

int pf_pack_theta(const pf_problem* p, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  PF_TRY(pf_launch_pack_theta(p, (hipStream_t)stream), "pf_pack_theta");
  return PF_OK;
}

int pf_net_forward(const pf_problem* p, int which, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  if (which < 0 || which > 1 || !p->net[which].enabled) return fail(PF_ERR_ARG, "net not enabled");
  PF_TRY(net_forward(p, which, (hipStream_t)stream), "pf_net_forward");
  return PF_OK;
}

int pf_net_forward_all(const pf_problem* p, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  PF_TRY(net_forward_all(p, (hipStream_t)stream), "pf_net_forward_all");
  return PF_OK;
}

int pf_net_backward_all(const pf_problem* p, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (!(p->net[0].enabled || p->net[1].enabled)) return PF_OK;
  if (can_fuse_backward(p)) {
    PF_TRY(net_backward2(p, s), "net_backward2");
    return PF_OK;
  }
  const bool fuse_gea = fuse_gea_for(p);
  const int first = p->net[0].enabled ? 0 : 1;
  if (!fuse_gea) PF_TRY(pf_launch_elem_adjoint(p, s), "elem_adjoint");
  for (int k = 0; k < 2; ++k)
    if (p->net[k].enabled)
      PF_TRY(fuse_gea && k == first ? net_backward_gea(p, k, s) : net_backward(p, k, s), "net_backward");
  return PF_OK;
}

int pf_internal_force(const pf_problem* p, const float* u, float* f_int_out, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  if (!u || !f_int_out) return fail(PF_ERR_ARG, "null u / f_int_out");
  pf_problem q = *p;
  q.u = const_cast<float*>(u);
  PF_TRY(pf_launch_node_residual(&q, f_int_out, 0, (hipStream_t)stream), "pf_internal_force");
  return PF_OK;
}

int pf_node_residual(const pf_problem* p, float* f_int_out, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  PF_TRY(pf_launch_node_residual(p, f_int_out, 1, (hipStream_t)stream), "pf_node_residual");
  return PF_OK;
}

int pf_elem_adjoint(const pf_problem* p, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  if (!p->g_ea) return fail(PF_ERR_ARG, "null g_ea");
  PF_TRY(pf_launch_elem_adjoint(p, (hipStream_t)stream), "pf_elem_adjoint");
  return PF_OK;
}

int pf_net_backward(const pf_problem* p, int which, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  if (which < 0 || which > 1 || !p->net[which].enabled) return fail(PF_ERR_ARG, "net not enabled");
  PF_TRY(net_backward(p, which, (hipStream_t)stream), "pf_net_backward");
  return PF_OK;
}

int pf_node_gradu(const pf_problem* p, int fuse_adam, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  if (fuse_adam && (!p->m_u || !p->v_u)) return fail(PF_ERR_ARG, "null Adam moments for u");
  if (!fuse_adam && !p->grad_u) return fail(PF_ERR_ARG, "grad_u required when Adam is not fused");
  PF_TRY(pf_launch_node_gradu(p, fuse_adam, (hipStream_t)stream), "pf_node_gradu");
  return PF_OK;
}

int pf_theta_reduce(const pf_problem* p, int fuse_adam, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  if (fuse_adam && p->n_theta_active > 0 && (!p->m_t || !p->v_t))
    return fail(PF_ERR_ARG, "null Adam moments for theta");
  PF_TRY(pf_launch_theta_reduce(p, fuse_adam, (hipStream_t)stream), "pf_theta_reduce");
  return PF_OK;
}

int pf_finalize(const pf_problem* p, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  PF_TRY(pf_launch_finalize(p, 0, 0, (hipStream_t)stream), "pf_finalize");
  return PF_OK;
}

int pf_reset(const pf_problem* p, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  if (!p->m_u || !p->v_u) return fail(PF_ERR_ARG, "null Adam moments for u");
  if (p->n_theta > 0 && (!p->m_t || !p->v_t)) return fail(PF_ERR_ARG, "null Adam moments for theta");
  PF_TRY(pf_launch_reset(p, (hipStream_t)stream), "pf_reset");
  return PF_OK;
}

// kernel slots of one iteration, in launch order (also the index into pf_gd_iterations_timed's output)
enum { K_FWD_E = 0, K_FWD_A, K_RESIDUAL, K_ADJOINT, K_BWD_E, K_BWD_A, K_GRADU, K_THETA, K_FINALIZE, K_COUNT };

// ev: optional array of K_COUNT+1 events; ev[k] is recorded before slot k, ev[K_COUNT] at the end
static int enqueue_iteration(const pf_problem* p, int fuse_adam, int finalize_mode, hipStream_t s,
                             hipEvent_t* ev) {
  const bool any_net = p->net[0].enabled || p->net[1].enabled;
#define PF_MARK(k) do { if (ev && hipEventRecord(ev[k], s) != hipSuccess) return fail(PF_ERR_HIP, "hipEventRecord failed"); } while (0)
  PF_MARK(K_FWD_E);
  const bool fwd2 = can_fuse_forward(p);              // both nets in the first slot's launch; the second slot stays empty
  if (fwd2) PF_TRY(net_forward2(p, s), "net_forward2");
  else if (p->net[0].enabled) PF_TRY(net_forward(p, 0, s, !p->net[1].enabled), "net_forward");
  PF_MARK(K_FWD_A);
  if (!fwd2 && p->net[1].enabled) PF_TRY(net_forward(p, 1, s), "net_forward");
  PF_MARK(K_RESIDUAL);
  PF_TRY(pf_launch_node_residual(p, nullptr, 1, s), "node_residual");
  const bool fuse_gea = any_net && fuse_gea_for(p);
  const int first = p->net[0].enabled ? 0 : 1;
  PF_MARK(K_ADJOINT);
  if (any_net && !fuse_gea) PF_TRY(pf_launch_elem_adjoint(p, s), "elem_adjoint");
  PF_MARK(K_BWD_E);
  const bool bwd2 = can_fuse_backward(p);             // both nets in the first slot's launch; the second slot stays empty
  if (bwd2) PF_TRY(net_backward2(p, s), "net_backward2");
  else if (p->net[0].enabled)
    PF_TRY(fuse_gea && first == 0 ? net_backward_gea(p, 0, s) : net_backward(p, 0, s), "net_backward");
  PF_MARK(K_BWD_A);
  if (!bwd2 && p->net[1].enabled)
    PF_TRY(fuse_gea && first == 1 ? net_backward_gea(p, 1, s) : net_backward(p, 1, s), "net_backward");
  PF_MARK(K_GRADU);
  PF_TRY(pf_launch_node_gradu(p, fuse_adam, s), "node_gradu");
  PF_MARK(K_THETA);
  if (any_net) PF_TRY(pf_launch_theta_stage1(p, s), "theta_stage1");
  PF_MARK(K_FINALIZE);
  PF_TRY(pf_launch_finalize(p, finalize_mode, any_net ? 1 : 0, s), "finalize");
  PF_MARK(K_COUNT);
#undef PF_MARK
  return PF_OK;
}

static int check_gd(const pf_problem* p) {
  int rc = check_problem(p);
  if (rc) return rc;
  if (!p->m_u || !p->v_u) return fail(PF_ERR_ARG, "null Adam moments for u");
  if (p->n_theta_active > 0 && (!p->m_t || !p->v_t)) return fail(PF_ERR_ARG, "null Adam moments for theta");
  if (p->n_tensors > 64) return fail(PF_ERR_UNSUPPORTED, "more than 64 parameter tensors");   // PF_MAX_TENSORS (pf_mesh.hip)
  return PF_OK;
}

int pf_gd_iterations(const pf_problem* p, int n_iter, void* stream) {
  int rc = check_gd(p);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < n_iter; ++i) {
    rc = enqueue_iteration(p, 1, 0, s, nullptr);
    if (rc) return rc;
  }
  return PF_OK;
}

// Iterations as a dependency graph instead of a chain (only meaningful while capturing a hipGraph).
// What each kernel of iteration t really waits for:
//   forward (E and A in one launch)   the second-level gradient rows of t-1 [theta stage 1 of t-1]: its prologue IS the
//                       parameter update of t-1 (every block for itself; MFMA32 engine);  nothing reads the properties
//                       or stiffness records of the other half any more  [gradu of t-1]
//   node_residual       forward, u(t) [gradu of t-1]; its block 0 = finalize(t-1): the update of t-1, gradu of t-1, and the
//                       OTHER half of the residual's partial sums (part_half)
//   backward #1 (+gea)  g_f; it is the last reader of u(t)
//   gradu + Adam(u)     g_f, the stiffness records, backward #1 done; the Adam scalars          [finalize of t-1]
//   backward #2, theta stage 1   in this order after backward #1 (the last iteration of a replay: + the stand-alone update)
//   finalize(t)         the update and gradu of t: it runs inside node_residual(t+1)
// so gradu (HBM bound) runs on branch A beside backward #2 and the theta reduction (compute bound).  The stop flag
// is read by every kernel at its start; a kernel of t+1 that misses a stop raised by finalize(t) only rewrites scratch
// (properties, g_f, partial sums): everything that changes solver state (both Adam kernels, the next finalize)
// is ordered behind finalize(t) and returns at once, so the final state is the reference's `break`.
#define PF_CAP_EV 6
// below this many elements the kernels are too short to hide anything behind: the branches' fork/join cost
// (~5 us each against ~1.5 us for a plain boundary) would only add to a launch-bound iteration
struct pf_capture {
  hipStream_t s, a, b;
  hipEvent_t* ev;   // PF_CAP_EV per iteration: u readers done | gradu done | theta done | finalize done | forward fork | join
};

static int cap_edge(hipEvent_t e, hipStream_t from, hipStream_t to) {
  if (hipEventRecord(e, from) != hipSuccess || hipStreamWaitEvent(to, e, 0) != hipSuccess)
    return fail(PF_ERR_HIP, "graph edge failed");
  return PF_OK;
}

// Does pf_problem.pad_index hold exactly what pf_pad_index_of computes (the layout pinn_fem_amd builds and INTEGRATION.md
// describes: nets in order, pf_net_pad_index + pad_off)?  Then the forward launch's update prologue computes the index
// instead of loading it.  One synchronous copy of the table, at graph creation only.
static bool pad_index_is_canonical(const pf_problem* p) {
  const int n = p->n_theta_active;
  if (n <= 0 || !p->pad_index) return false;
  int* h = new int[n];
  const bool ok = hipMemcpy(h, p->pad_index, (size_t)n * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
  bool same = ok;
  for (int q = 0; same && q < n; ++q) same = h[q] == pf_pad_index_of(*p, q);
  delete[] h;
  return same;
}

// head_cont: iteration 0 also carries the updates and the bookkeeping of the iteration BEFORE the replay (left pending by a
// no_tail replay); no_tail: the replay ends behind the last iteration's gradient-row reduction, its parameter update,
// displacement update and bookkeeping stay pending (pf_graph_create_ex).  Both only in the one-chain form.
static int enqueue_graph_iterations(const pf_problem* p, int iters, const pf_capture& c, bool calc_index = false,
                                    bool head_cont = false, bool no_tail = false) {
  const bool any_net = p->net[0].enabled || p->net[1].enabled;
  hipStream_t s = c.s;
  // PF_GRAPH_SERIAL=1: experiment knob, the plain chain of launches inside the graph (no branches)
  static const int serial_knob = getenv("PF_GRAPH_SERIAL") ? atoi(getenv("PF_GRAPH_SERIAL")) : -1;
  bool serial = serial_knob >= 0 ? serial_knob != 0 : p->mesh.n_elems < PF_GRAPH_DAG_MIN_ELEMS;
  if (!any_net) {
    for (int i = 0; i < iters; ++i) {
      int rc = enqueue_iteration(p, 1, 0, s, nullptr);
      if (rc != PF_OK) return rc;
    }
    return PF_OK;
  }
  // serial: the same kernels as ONE chain on `s` (no side branch, no events): forward, residual [+ finalize of the
  // previous iteration in its block 0], backwards, theta stage 1, gradu — neither the bookkeeping nor the parameter update
  // costs a launch of its own there either.
  static const bool pp_knob = !(getenv("PF_GRAPH_PINGPONG") && atoi(getenv("PF_GRAPH_PINGPONG")) == 0);
  const bool pingpong = p->prop_double != 0 && pp_knob;
  // The displacement update of iteration i-1 (dL/du + Adam(u) + clamp) runs INSIDE iteration i's forward launch, as node
  // tasks between its element tasks (can_fuse_gradu; it reads the stiffness records of i-1 = the other half, which is
  // why it needs the two halves): the graph is then ONE chain.  Only the replay's last iteration keeps the stand-alone
  // kernel, in place.
  const bool fuse_gu = pingpong && p->elem_k != nullptr && can_fuse_gradu(p);
  if (fuse_gu) serial = true;
  hipStream_t sa = serial ? s : c.a;
  auto ev_wait = [&](hipStream_t st, hipEvent_t e) { return serial || hipStreamWaitEvent(st, e, 0) == hipSuccess; };
  auto ev_rec = [&](hipEvent_t e, hipStream_t st) { return serial || hipEventRecord(e, st) == hipSuccess; };
  const bool fuse_gea = fuse_gea_for(p);
  const int first = p->net[0].enabled ? 0 : 1;
  // The runtime keeps the FIRST-created child of a node on its parent's hardware queue and moves later
  // children to other queues; every fork or join on the main chain costs ~5 us (kernel trace).  So the nodes are
  // created in the order that keeps the critical chain (forwards, residual, backwards, theta) on one queue, and
  // the bookkeeping kernel has NO node of its own: finalize(i-1) runs as block 0 of node_residual(i)
  // (pf_mesh.hip: k_node_residual, fin_prev) from the other half of the residual's partial sums
  // (pf_problem.part_half), finalize of the graph's last iteration as a stand-alone launch at its end.
  const int tn_ready = p->wg_mode == PF_WG_MFMA32 ? 1 : 0;   // k_theta_stage2 leaves the theta-norm monitor in the state
  // The parameter update of iteration i-1 (theta stage 2: second-level rows -> Adam -> operand images) runs in the
  // PROLOGUE of iteration i's forward launch, by every block for itself (pf_net32.hip: fwd_theta_prologue); the state
  // ping-pongs between its two halves so that block 0's stores never meet another block's loads.  Only the replay's
  // last iteration keeps the stand-alone update, which also brings the state back to half 0.
  const bool fuse_s2 = can_fuse_theta_update(p);
  // Displacement vectors ping-pong too (pf_problem.u_alt; DAG form, even replay length): the update of iteration i reads
  // U[i & 1] like the residual and the element adjoint of i and WRITES U[(i + 1) & 1], so it forks right behind the
  // residual and runs beside the whole backward launch.  PF_GRAPH_UPP=0: experiment knob (in place, fork behind the
  // adjoint's last read of u).
  static const bool upp_knob = !(getenv("PF_GRAPH_UPP") && atoi(getenv("PF_GRAPH_UPP")) == 0);
  const bool upp = !serial && upp_knob && p->u_alt != nullptr && (iters % 2) == 0;
  const float* c_elem_k = p->elem_k;
  for (int i = 0; i < iters; ++i) {
    hipEvent_t* e = c.ev + PF_CAP_EV * i;
    hipEvent_t* ep = c.ev + PF_CAP_EV * (i - 1);
    // Property buffers ping-pong between iterations (prop_double), so the forwards of iteration i do not wait
    // for gradu(i-1), the last reader of the other half: the main chain then has ONE incoming edge from another
    // queue per iteration (gradu(i-1) -> residual(i): u, and the u-norm partials finalize(i-1) reads there).
    // Without the second half the forwards wait for gradu(i-1) (an 11 us hole in the kernel trace).
    pf_problem q = *p;
    const pf_problem* p = &q;   // (shadows the argument for the rest of this iteration)
    if (pingpong && (i & 1)) {
      q.prop_e += q.mesh.n_elems;
      q.prop_a += q.mesh.n_elems;
      if (q.elem_k) q.elem_k += (size_t)q.mesh.n_elems * (q.mesh.dim == 2 ? 3 : 1);
    }
    q.part_half = i & 1;
    float* u_next = nullptr;
    if (upp) {
      u_next = (i & 1) ? q.u : q.u_alt;
      if (i & 1) q.u = q.u_alt;                       // what this iteration's kernels READ
    }
    if (i > 0 && !pingpong && !ev_wait(s, ep[1])) return fail(PF_ERR_HIP, "graph edge failed");
    // (both nets in one launch where the engine has it; else one after the other: side by side on two branches they
    // measured slower, the same issue pipe, and the second one writes the stiffness records from both)
    const bool carries = i > 0 || head_cont;      // this iteration's forward / residual carry the previous iteration's work
    const float* k_prev = nullptr;        // the records iteration i-1 wrote (the other half)
    if (fuse_gu && carries) k_prev = ((i - 1) & 1) ? c_elem_k + (size_t)q.mesh.n_elems * (q.mesh.dim == 2 ? 3 : 1) : c_elem_k;
    pf_fwd2_opts fo;
    if (fuse_s2 && carries) fo.s2_half = (i - 1) & 1;       // (i = 0 in a continuation: the previous replay's last = half 1)
    fo.calc_index = calc_index;
    if (k_prev) {
      fo.gu_nb = pf_node_blocks(q.mesh.n_nodes);
      fo.gu_k = k_prev;
    }
    PF_TRY(net_forward_all(p, s, fo), "net_forward");
    // residual(i) reads u(i) [gradu(i-1)]; its block 0 is finalize(i-1): behind the theta update of i-1 (this chain:
    // the forward launch above, or the stand-alone kernel) and gradu(i-1)
    if (i > 0 && !ev_wait(s, ep[1])) return fail(PF_ERR_HIP, "graph edge failed");
    PF_TRY(pf_launch_node_residual(p, nullptr, 1, s, carries ? (tn_ready ? 2 : 1) : 0), "node_residual");
    if (upp && !ev_rec(e[0], s)) return fail(PF_ERR_HIP, "graph edge failed");      // gradu forks here
    if (!fuse_gea) {
      PF_TRY(pf_launch_elem_adjoint(p, s), "elem_adjoint");
      if (!upp && !ev_rec(e[0], s)) return fail(PF_ERR_HIP, "graph edge failed");
    }
    if (can_fuse_backward(p)) {
      // both backward passes in one launch (without the second displacement vector gradu can only fork behind it)
      PF_TRY(net_backward2(p, s, fuse_s1_knob() ? 1 : 0), "net_backward2");
      if (!upp && !ev_rec(e[0], s)) return fail(PF_ERR_HIP, "graph edge failed");
    } else {
      for (int k = 0; k < 2; ++k) {
        if (!p->net[k].enabled) continue;
        PF_TRY(fuse_gea && k == first ? net_backward_gea(p, k, s) : net_backward(p, k, s), "net_backward");
        if (!upp && fuse_gea && k == first && !ev_rec(e[0], s)) return fail(PF_ERR_HIP, "graph edge failed");
      }
    }
    if (!(can_fuse_backward(p) && fuse_s1_knob())) PF_TRY(pf_launch_theta_stage1(p, s), "theta_stage1");
    if (!fuse_s2 || (i == iters - 1 && !no_tail)) PF_TRY(pf_launch_theta_stage2(p, 1, s), "theta_stage2");
    // branch A (created after the main chain's nodes of this iteration): gradu behind the last reader of u
    if (fuse_gu && (i < iters - 1 || no_tail)) continue;   // (the next forward launch carries it)
    if (!ev_wait(sa, e[0])) return fail(PF_ERR_HIP, "graph edge failed");
    PF_TRY(pf_launch_node_gradu(p, 1, sa, 0, u_next), "node_gradu");
    if (!ev_rec(e[1], sa)) return fail(PF_ERR_HIP, "graph edge failed");
  }
  // finalize of the last iteration: behind stage 2 (this chain) and the last gradu
  if (no_tail) return PF_OK;
  if (!serial && !ev_wait(s, c.ev[PF_CAP_EV * (iters - 1) + 1])) return fail(PF_ERR_HIP, "graph join failed");
  if (upp) PF_TRY(pf_launch_u_home(p, s), "u_home");      // (only a stop in mid-replay leaves anything to copy)
  {
    pf_problem q = *p;
    q.part_half = (iters - 1) & 1;
    PF_TRY(pf_launch_finalize(&q, 0, 0, s, tn_ready), "finalize");
  }
  return PF_OK;
}

}  // extern "C"
// capture fn(capture streams/events) on `s` (+ two side streams) and instantiate the graph
template <class F>
static int capture_graph(hipStream_t s, int nev, hipStreamCaptureMode mode, void** graph_out, F&& fn) {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipStream_t side[2] = {nullptr, nullptr};
  hipEvent_t* ev = new hipEvent_t[nev];
  int made = 0;
  // PF_GRAPH_SIDE_PRIO (experiment knob): priority of the capture's side streams (-1 high, 0 default, 1 low)
  static const int prio_knob = getenv("PF_GRAPH_SIDE_PRIO") ? atoi(getenv("PF_GRAPH_SIDE_PRIO")) : 0;
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  const int prio = prio_knob < 0 ? hi : (prio_knob > 0 ? lo : 0);
  bool ok = hipStreamCreateWithPriority(&side[0], hipStreamNonBlocking, prio) == hipSuccess &&
            hipStreamCreateWithPriority(&side[1], hipStreamNonBlocking, prio) == hipSuccess;
  for (; ok && made < nev; ++made)
    if (hipEventCreateWithFlags(&ev[made], hipEventDisableTiming) != hipSuccess) break;
  ok = ok && made == nev;
  auto cleanup = [&]() {
    for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
    delete[] ev;
    for (int k = 0; k < 2; ++k)
      if (side[k]) (void)hipStreamDestroy(side[k]);
  };
  if (!ok) {
    cleanup();
    return fail(PF_ERR_HIP, "graph capture: stream/event creation failed");
  }
  if (hipStreamBeginCapture(s, mode) != hipSuccess) {
    cleanup();
    return fail(PF_ERR_HIP, "hipStreamBeginCapture failed");
  }
  pf_capture cap{s, side[0], side[1], ev};
  const int rc = fn(cap);
  const hipError_t e = hipStreamEndCapture(s, &graph);
  cleanup();
  if (rc != PF_OK) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  if (e != hipSuccess || !graph) return fail(PF_ERR_HIP, "hipStreamEndCapture failed");
  if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
    (void)hipGraphDestroy(graph);
    return fail(PF_ERR_HIP, "hipGraphInstantiate failed");
  }
  (void)hipGraphDestroy(graph);
  *graph_out = (void*)exec;
  return PF_OK;
}
extern "C" {

int pf_graph_create(const pf_problem* p, int iters_per_graph, void* stream, void** graph_out) {
  int rc = check_gd(p);
  if (rc) return rc;
  if (!graph_out || iters_per_graph < 1) return fail(PF_ERR_ARG, "pf_graph_create: bad argument");
  static const int ci_knob = getenv("PF_CALC_INDEX") ? atoi(getenv("PF_CALC_INDEX")) : 1;   // 0: experiment knob (index table)
  const bool calc_index = ci_knob != 0 && can_fuse_theta_update(p) && pad_index_is_canonical(p);
  return capture_graph((hipStream_t)stream, PF_CAP_EV * iters_per_graph, hipStreamCaptureModeThreadLocal, graph_out,
                       [&](pf_capture& cap) { return enqueue_graph_iterations(p, iters_per_graph, cap, calc_index); });
}

// can the iteration graph of this problem be cut into replays that hand their last iteration's updates to the next one?
static bool can_chain_replays(const pf_problem* p, int iters) {
  static const bool pp_knob = !(getenv("PF_GRAPH_PINGPONG") && atoi(getenv("PF_GRAPH_PINGPONG")) == 0);
  return (iters % 2) == 0 && p->prop_double != 0 && pp_knob && p->elem_k != nullptr && can_fuse_gradu(p) &&
         can_fuse_theta_update(p);
}

int pf_graph_create_ex(const pf_problem* p, int iters_per_graph, int flags, void* stream, void** graph_out) {
  int rc = check_gd(p);
  if (rc) return rc;
  if (!graph_out || iters_per_graph < 1) return fail(PF_ERR_ARG, "pf_graph_create_ex: bad argument");
  if ((flags & (PF_GRAPH_CONT_HEAD | PF_GRAPH_NO_TAIL)) && !can_chain_replays(p, iters_per_graph))
    return fail(PF_ERR_UNSUPPORTED, "pf_graph_create_ex: replays of this problem's graph cannot hand over their tail");
  static const int ci_knob = getenv("PF_CALC_INDEX") ? atoi(getenv("PF_CALC_INDEX")) : 1;
  const bool calc_index = ci_knob != 0 && can_fuse_theta_update(p) && pad_index_is_canonical(p);
  const bool head = (flags & PF_GRAPH_CONT_HEAD) != 0, no_tail = (flags & PF_GRAPH_NO_TAIL) != 0;
  return capture_graph((hipStream_t)stream, PF_CAP_EV * iters_per_graph, hipStreamCaptureModeThreadLocal, graph_out,
                       [&](pf_capture& cap) { return enqueue_graph_iterations(p, iters_per_graph, cap, calc_index, head, no_tail); });
}

// the pending tail of a PF_GRAPH_NO_TAIL replay: what the last iteration of a plain replay ends with
int pf_graph_tail(const pf_problem* p, int iters_per_graph, void* stream) {
  int rc = check_gd(p);
  if (rc) return rc;
  if (!can_chain_replays(p, iters_per_graph)) return fail(PF_ERR_UNSUPPORTED, "pf_graph_tail: not a chained replay");
  hipStream_t s = (hipStream_t)stream;
  pf_problem q = *p;                      // the last iteration's halves (iters even: the second ones)
  const int i = iters_per_graph - 1;
  if (i & 1) {
    q.prop_e += q.mesh.n_elems;
    q.prop_a += q.mesh.n_elems;
    if (q.elem_k) q.elem_k += (size_t)q.mesh.n_elems * (q.mesh.dim == 2 ? 3 : 1);
  }
  q.part_half = i & 1;
  PF_TRY(pf_launch_theta_stage2(&q, 1, s), "theta_stage2");
  PF_TRY(pf_launch_node_gradu(&q, 1, s, 0, nullptr), "node_gradu");
  PF_TRY(pf_launch_finalize(&q, 0, 0, s, q.wg_mode == PF_WG_MFMA32 ? 1 : 0), "finalize");
  return PF_OK;
}

int pf_graph_launch(void* graph, void* stream) {
  if (!graph) return fail(PF_ERR_ARG, "null graph");
  if (hipGraphLaunch((hipGraphExec_t)graph, (hipStream_t)stream) != hipSuccess)
    return fail(PF_ERR_HIP, "hipGraphLaunch failed");
  return PF_OK;
}

int pf_graph_destroy(void* graph) {
  if (!graph) return PF_OK;
  if (hipGraphExecDestroy((hipGraphExec_t)graph) != hipSuccess) return fail(PF_ERR_HIP, "hipGraphExecDestroy failed");
  return PF_OK;
}

int pf_gd_iterations_timed(const pf_problem* p, int n_iter, void* stream, float* ms_per_kernel) {
  int rc = check_gd(p);
  if (rc) return rc;
  if (!ms_per_kernel || n_iter < 1) return fail(PF_ERR_ARG, "pf_gd_iterations_timed: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const int per = K_COUNT + 1;
  hipEvent_t* ev = new hipEvent_t[(size_t)n_iter * per];
  int made = 0;
  for (; made < n_iter * per; ++made)
    if (hipEventCreate(&ev[made]) != hipSuccess) break;
  rc = made == n_iter * per ? PF_OK : fail(PF_ERR_HIP, "hipEventCreate failed");
  for (int i = 0; rc == PF_OK && i < n_iter; ++i) rc = enqueue_iteration(p, 1, 0, s, ev + (size_t)i * per);
  if (rc == PF_OK && hipStreamSynchronize(s) != hipSuccess) rc = fail(PF_ERR_HIP, "hipStreamSynchronize failed");
  if (rc == PF_OK) {
    for (int k = 0; k < K_COUNT; ++k) ms_per_kernel[k] = 0.f;
    for (int i = 0; i < n_iter; ++i)
      for (int k = 0; k < K_COUNT; ++k) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev[(size_t)i * per + k], ev[(size_t)i * per + k + 1]) != hipSuccess) {
          rc = fail(PF_ERR_HIP, "hipEventElapsedTime failed");
          break;
        }
        ms_per_kernel[k] += ms;
      }
    for (int k = 0; k < K_COUNT; ++k) ms_per_kernel[k] /= (float)n_iter;
  }
  for (int i = 0; i < made; ++i) hipEventDestroy(ev[i]);
  delete[] ev;
  return rc;
}

int pf_loss_and_grads(const pf_problem* p, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  if (!p->grad_u) return fail(PF_ERR_ARG, "grad_u required");
  return enqueue_iteration(p, 0, 1, (hipStream_t)stream, nullptr);
}

static int check_shared(const pf_problem* p) {
  int rc = check_problem(p);
  if (rc) return rc;
  if (p->n_shared < 0 || p->n_iface < p->n_shared) return fail(PF_ERR_ARG, "bad interface sizes");
  if (p->n_shared > 0 && (!p->shared_dofs || !p->shared_slot)) return fail(PF_ERR_ARG, "null interface maps");
  if (p->own_lo < 0 || p->own_hi < p->own_lo || p->own_hi > p->mesh.n_elems) return fail(PF_ERR_ARG, "bad own element range");
  return PF_OK;
}

// The own elements as a problem of their own: the element-indexed arrays start at own_lo (node ids in the
// connectivity are untouched), so the element-parallel kernels run on exactly the range whose gradients this rank owns.
static pf_problem own_view(const pf_problem* p) {
  pf_problem q = *p;
  if (p->own_hi > 0) {
    const int lo = p->own_lo;
    q.mesh.conn += 2 * (size_t)lo;
    q.mesh.egeo += 4 * (size_t)lo;
    q.mesh.ecent += (size_t)p->mesh.dim * lo;
    if (q.prop_e) q.prop_e += lo;
    if (q.prop_a) q.prop_a += lo;
    if (q.g_ea) q.g_ea += lo;
    if (q.elem_k) q.elem_k += (size_t)lo * (p->mesh.dim == 2 ? 3 : 1);
    q.mesh.n_elems = p->own_hi - lo;
  }
  return q;
}

int pf_shard_forward(const pf_problem* p, void* stream) {
  int rc = check_shared(p);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  PF_TRY(net_forward_all(p, s), "net_forward");
  return PF_OK;
}

int pf_shard_backward(const pf_problem* p, float* buf, const float* u2_local, void* stream) {
  int rc = check_shared(p);
  if (rc) return rc;
  if (!buf || !u2_local) return fail(PF_ERR_ARG, "null buf / u2_local");
  if (p->n_theta_active > 0 && p->grad_theta != buf + p->n_iface)
    return fail(PF_ERR_ARG, "p->grad_theta must point at buf + n_iface");
  hipStream_t s = (hipStream_t)stream;
  const bool any_net = p->net[0].enabled || p->net[1].enabled;
  PF_TRY(pf_launch_node_residual(p, nullptr, 1, s), "node_residual");
  if (any_net) {
    const pf_problem q = own_view(p);
    const bool fuse_gea = fuse_gea_for(&q);
    const int first = q.net[0].enabled ? 0 : 1;
    if (can_fuse_backward(&q)) {
      PF_TRY(net_backward2(&q, s), "net_backward2");
    } else {
      if (!fuse_gea) PF_TRY(pf_launch_elem_adjoint(&q, s), "elem_adjoint");
      for (int k = 0; k < 2; ++k)
        if (q.net[k].enabled)
          PF_TRY(fuse_gea && k == first ? net_backward_gea(&q, k, s) : net_backward(&q, k, s), "net_backward");
    }
    PF_TRY(pf_launch_theta_stage1(&q, s), "theta_stage1");
  }
  PF_TRY(pf_launch_shard_pack(p, buf, u2_local, s), "shard_pack");
  return PF_OK;
}

int pf_shard_update_interior(const pf_problem* p, void* stream) {
  int rc = check_shared(p);
  if (rc) return rc;
  if (!p->m_u || !p->v_u) return fail(PF_ERR_ARG, "null Adam moments for u");
  PF_TRY(pf_launch_node_gradu(p, 1, (hipStream_t)stream, 1), "node_gradu");
  return PF_OK;
}

}  // extern "C"
// `iters` complete sharded iterations as ONE hipGraph, the collective included (all_reduce: the caller's, captured on
// the main stream like a kernel).  Per iteration: forwards, residual, first backward -> [fork: gradu of the interior
// dofs] -> second backward, theta stage 1, pack -> all-reduce(buf) -> [join] -> interface update + bookkeeping; i.e.
// the single-engine iteration's shape with the collective on the chain and gradu hidden beside the second backward
// and the collective.  Internal (pf_comm.hip: pf_shard_graph_create).
int pf_shard_graph_capture(const pf_problem* p, int iters, float* buf, float* u2_local, hipStream_t stream,
                           int (*all_reduce)(void* ctx, float* buf, size_t n, hipStream_t s), void* ctx, void** graph_out) {
  int rc = check_shared(p);
  if (rc) return rc;
  if (!buf || !u2_local || !graph_out || !all_reduce || iters < 1 || !p->m_u || !p->v_u)
    return fail(PF_ERR_ARG, "pf_shard_graph_capture: bad argument");
  if (p->n_theta_active > 0 && (p->grad_theta != buf + p->n_iface || !p->m_t || !p->v_t))
    return fail(PF_ERR_ARG, "p->grad_theta must point at buf + n_iface (and the theta moments must exist)");
  const size_t n = (size_t)p->n_iface + (size_t)p->n_theta_active + 3;
  // relaxed capture mode: the collective library may call into the runtime while it enqueues
  return capture_graph(stream, 2 * iters, hipStreamCaptureModeRelaxed, graph_out, [&](pf_capture& c) -> int {
    hipStream_t s = c.s;
    const bool any_net = p->net[0].enabled || p->net[1].enabled;
    for (int i = 0; i < iters; ++i) {
      hipEvent_t* e = c.ev + 2 * i;
      PF_TRY(net_forward_all(p, s), "net_forward");
      PF_TRY(pf_launch_node_residual(p, nullptr, 1, s), "node_residual");
      bool marked = false;                  // e[0]: the last reader of u (the element adjoint) is done
      auto mark = [&]() {
        marked = true;
        return hipEventRecord(e[0], s) == hipSuccess;
      };
      if (any_net) {
        const pf_problem q = own_view(p);
        const bool fuse_gea = fuse_gea_for(&q);
        const int first = q.net[0].enabled ? 0 : 1;
        if (can_fuse_backward(&q)) {
          PF_TRY(net_backward2(&q, s), "net_backward2");
          if (!mark()) return fail(PF_ERR_HIP, "graph edge failed");
        } else {
          if (!fuse_gea) {
            PF_TRY(pf_launch_elem_adjoint(&q, s), "elem_adjoint");
            if (!mark()) return fail(PF_ERR_HIP, "graph edge failed");
          }
          for (int k = 0; k < 2; ++k) {
            if (!q.net[k].enabled) continue;
            PF_TRY(fuse_gea && k == first ? net_backward_gea(&q, k, s) : net_backward(&q, k, s), "net_backward");
            if (fuse_gea && k == first && !mark()) return fail(PF_ERR_HIP, "graph edge failed");
          }
        }
        PF_TRY(pf_launch_theta_stage1(&q, s), "theta_stage1");
      }
      if (!marked && !mark()) return fail(PF_ERR_HIP, "graph edge failed");
      PF_TRY(pf_launch_shard_pack(p, buf, u2_local, s), "shard_pack");
      // the side branch is created AFTER the chain's nodes of this iteration: the runtime keeps the first-created child of
      // a node on its parent's hardware queue, and the chain must be the one that stays (pf_graph_create, same rule)
      if (hipStreamWaitEvent(c.a, e[0], 0) != hipSuccess) return fail(PF_ERR_HIP, "graph edge failed");
      PF_TRY(pf_launch_node_gradu(p, 1, c.a, 1), "node_gradu");
      if (hipEventRecord(e[1], c.a) != hipSuccess) return fail(PF_ERR_HIP, "graph edge failed");
      int r3 = all_reduce(ctx, buf, n, s);
      if (r3 != PF_OK) return r3;
      if (hipStreamWaitEvent(s, e[1], 0) != hipSuccess) return fail(PF_ERR_HIP, "graph join failed");
      PF_TRY(pf_launch_shard_update(p, buf, u2_local, s), "shard_update");
    }
    return PF_OK;
  });
}
extern "C" {

int pf_shard_update_shared(const pf_problem* p, const float* buf, float* u2_local, void* stream) {
  int rc = check_shared(p);
  if (rc) return rc;
  if (!buf || !u2_local || !p->m_u || !p->v_u) return fail(PF_ERR_ARG, "null buffer");
  if (p->n_theta_active > 0 && (!p->m_t || !p->v_t)) return fail(PF_ERR_ARG, "null Adam moments for theta");
  hipStream_t s = (hipStream_t)stream;
  PF_TRY(pf_launch_shard_update(p, buf, u2_local, s), "shard_update");   // (includes the iteration's bookkeeping)
  return PF_OK;
}

int pf_shard_flush(const pf_problem* p, const float* u2_reduced, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  if (!u2_reduced) return fail(PF_ERR_ARG, "null u2");
  PF_TRY(pf_launch_shard_flush(p, u2_reduced, (hipStream_t)stream), "shard_flush");
  return PF_OK;
}

}  // extern "C"
extern "C" {

int pf_scalar_gd_iterations(const pf_problem* p, const pf_scalar_id* sp, int n_iter, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  if (!sp || !sp->p || !sp->m_p || !sp->v_p || n_iter < 0 || (sp->n_rows > 0 && !sp->table))
    return fail(PF_ERR_ARG, "pf_scalar_gd_iterations: bad argument");
  if (p->net[0].enabled || p->net[1].enabled) return fail(PF_ERR_ARG, "pf_scalar_gd_iterations: scalar materials only");
  if (!p->m_u || !p->v_u || !(sp->n_free_f > 0.f)) return fail(PF_ERR_ARG, "pf_scalar_gd_iterations: missing state");
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < n_iter; ++i) {
    PF_TRY(pf_launch_scalar_residual(p, sp, s), "scalar_residual");
    PF_TRY(pf_launch_node_gradu(p, 1, s), "node_gradu");
    PF_TRY(pf_launch_scalar_update(p, sp, s), "scalar_update");
  }
  return PF_OK;
}

int pf_adam(float* param, const float* grad, float* m, float* v, int n, int step, double lr,
            double beta1, double beta2, double eps, void* stream) {
  if (!param || !grad || !m || !v || n < 0 || step < 1) return fail(PF_ERR_ARG, "pf_adam: bad argument");
  PF_TRY(pf_launch_adam(param, grad, m, v, n, step, lr, beta1, beta2, eps, (hipStream_t)stream), "pf_adam");
  return PF_OK;
}

int pf_diag_k(const pf_problem* p, float* diag_out, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  if (!diag_out) return fail(PF_ERR_ARG, "null diag_out");
  PF_TRY(pf_launch_diag_k(p, diag_out, (hipStream_t)stream), "pf_diag_k");
  return PF_OK;
}

int pf_coo_k(const pf_problem* p, long long* rows_out, long long* cols_out, float* vals_out, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  if (!rows_out || !cols_out || !vals_out) return fail(PF_ERR_ARG, "null COO output");
  PF_TRY(pf_launch_coo_k(p, rows_out, cols_out, vals_out, (hipStream_t)stream), "pf_coo_k");
  return PF_OK;
}

int pf_dense_k(const pf_problem* p, float* k_out, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  if (!k_out) return fail(PF_ERR_ARG, "null k_out");
  if (p->mesh.n_dofs > 4096) return fail(PF_ERR_ARG, "dense K is a small-problem view (n_dofs <= 4096)");
  PF_TRY(pf_launch_dense_k(p, k_out, (hipStream_t)stream), "pf_dense_k");
  return PF_OK;
}

}  // extern "C"
